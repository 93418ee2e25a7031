"""TEST INFRASTRUCTURE (CPU oracle, numpy) -- never imported by the product path.

Restatement of the tensor side of the reference's input pipeline (the part after JPEG decode / resize):
apps/eval.py:59-61 == dataset/interhand.py:223-225:
    imgTensor = torch.tensor(cv.cvtColor(img, cv.COLOR_BGR2RGB), dtype=torch.float32) / 255
    imgTensor = imgTensor.permute(2, 0, 1)
    imgTensor = Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])(imgTensor)      # (t - mean) / std
Pinned by tests/golden/g11_imgprep.npz (oracle/gen_golden.py::gen_imgprep executes those three reference statements).
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], np.float32)       # apps/eval.py:49-50
STD = np.array([0.229, 0.224, 0.225], np.float32)


def normalize_u8_bgr(img_u8_hwc_bgr):
    """[B,H,W,3] uint8 BGR -> [B,3,H,W] float32, same fp32 operation order as the reference (divide, subtract, divide)"""
    rgb = img_u8_hwc_bgr[..., ::-1].astype(np.float32) / np.float32(255)
    t = np.ascontiguousarray(rgb.transpose(0, 3, 1, 2))
    return ((t - MEAN[None, :, None, None]) / STD[None, :, None, None]).astype(np.float32)
