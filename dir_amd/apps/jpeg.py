"""From files at device rate: cv.imread of the reference's input pipeline (apps/eval.py:56,
dataset/interhand.py:223 over dataset/prepare_data.py:123-166's <split>/img/<idx>.jpg) split in two.

    host   (decode worker processes, lib/libdir_jpeg.so, include/dir_jpeg.h)   the Huffman stream -> one RECORD per image: a 512-byte header + the
                                                                               quantised DCT coefficients (int16) -- the only inherently serial part
    GPU    (dir_jpeg_decode_records, csrc/jpeg.hip)                            dequantise + integer IDCT + chroma upsampling + YCbCr -> BGR uint8
                                                                               [B,256,256,3]: what the stem kernel reads

bit-exact with libjpeg(-turbo), the decoder OpenCV and Pillow both use (oracle/jpeg.py is pinned to Pillow's).  A 4:2:0 record has the size of
the decoded frame, so the host -> device traffic is unchanged while the host's work per image falls to the entropy decode (PIL's full decode was
what saturated the 16-CPU quota at 22 k images/s, profiles/r04_fromdisk_sweep_flag_ring.txt).  Files the entropy decoder refuses (progressive,
arithmetic, CMYK) or that are not 256x256 are decoded the ordinary way on the host (dataset.decode_bgr) and travel as PIXEL records.

Plumbing only: pointers, sizes, error codes.  No GPU call happens in a worker process (libdir_jpeg.so has no GPU runtime in it)."""
import ctypes as C
import os

import numpy as np
import torch

from .. import _capi

HEADER_BYTES = 512
MAGIC, MAGIC_PIXELS = 0x4a524944, 0x50524944
_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.environ.get('DIR_JPEG_LIB_PATH') or os.path.join(os.path.dirname(_HERE), 'lib', 'libdir_jpeg.so')
_host = None


def host_lib():
    """libdir_jpeg.so (built by dir_amd/build.py with gcc).  No fallback: a missing library is an error."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise _capi.DirHipError('%s not found: run python -m dir_amd.build' % HOST_LIB_PATH)
        lib = C.CDLL(HOST_LIB_PATH)
        lib.dir_jpeg_decode_coefficients.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.dir_jpeg_decode_coefficients.restype = C.c_int
        lib.dir_jpeg_record_bytes.argtypes = [C.c_int] * 5
        lib.dir_jpeg_record_bytes.restype = C.c_size_t
        lib.dir_jpeg_abi_version.restype = C.c_int
        if lib.dir_jpeg_abi_version() != 1:
            raise _capi.DirHipError('libdir_jpeg.so ABI version %d != 1' % lib.dir_jpeg_abi_version())
        _host = lib
    return _host


def record_bytes(size=256, hsamp=2, vsamp=2, ncomp=3):
    """bytes of one record of a size x size image with these luma sampling factors (4:2:0 = the default of cv.imwrite / libjpeg), a multiple of 16"""
    n = int(host_lib().dir_jpeg_record_bytes(size, size, hsamp, vsamp, ncomp))
    assert n > 0
    return (n + 15) // 16 * 16


def file_to_record(path, row, size=256):
    """one file -> `row` (a contiguous uint8 numpy array of record size).  Baseline size x size JPEGs become coefficient records; anything else is
    decoded on the host (dataset.decode_bgr: libjpeg through PIL, resized like cv.resize) and becomes a pixel record.  Returns 'coef' | 'pixels'."""
    with open(path, 'rb') as f:
        data = f.read()
    rc = host_lib().dir_jpeg_decode_coefficients(data, len(data), row.ctypes.data, row.size)
    if rc == 0:
        hdr = row[:16].view(np.int32)
        if hdr[1] == size and hdr[2] == size:
            return 'coef'
    # -3 unsupported coding / -4 more coefficients than the ring's records hold (4:4:4 in a 4:2:0 ring) / another size: the host decodes this one
    if rc in (-1, -2):
        raise ValueError('%s: not a decodable JPEG (dir_jpeg_decode_coefficients returned %d)' % (path, rc))
    from .dataset import decode_bgr
    px = decode_bgr(path, size)
    need = HEADER_BYTES + px.size
    if row.size < need:
        raise ValueError('%s: a %d-byte record cannot hold the decoded %dx%d frame' % (path, row.size, size, size))
    row[:HEADER_BYTES] = 0
    hdr = row[:16].view(np.int32)
    hdr[0], hdr[1], hdr[2], hdr[3] = MAGIC_PIXELS, size, size, 3
    row[HEADER_BYTES:need] = px.reshape(-1)
    return 'pixels'


class RecordDecoder(object):
    """records on the GPU -> uint8 BGR frames [B,size,size,3] (dir_jpeg_decode_records), with its scratch planes and error word kept per instance"""

    def __init__(self, batch, stride, size=256, device='cuda'):
        """size: the frames' edge, or (height, width)"""
        self.batch, self.stride = batch, stride
        self.H, self.W = (size, size) if isinstance(size, int) else size
        L = _capi.lib()
        per = (int(L.dir_jpeg_planes_bytes(stride)) + 15) // 16 * 16
        self.scratch = torch.empty(batch * per, dtype=torch.uint8, device=device)
        self.err = torch.zeros(1, dtype=torch.int32, device=device)

    def __call__(self, records, out, n=None):
        """records: uint8 cuda [B, stride]; out: uint8 cuda [B,size,size,3]; decodes the first n (default all) on the current stream"""
        _capi.require_cuda(records, out)
        assert records.dtype == torch.uint8 and records.is_contiguous() and records.shape[1] == self.stride and out.is_contiguous()
        n = records.shape[0] if n is None else n
        assert n <= self.batch and out.shape[0] >= n and tuple(out.shape[1:]) == (self.H, self.W, 3)
        with torch.cuda.device(records.device):
            _capi.check(_capi.lib().dir_jpeg_decode_records(_capi.ptr(records), self.stride, n, self.H, self.W, _capi.ptr(self.scratch), self.scratch.numel(),
                                                            _capi.ptr(out), _capi.ptr(self.err), _capi.stream_ptr()), 'dir_jpeg_decode_records')
        return out

    def check(self):
        """host read of the error word (a synchronisation: call it once per pass, not per batch)"""
        e = int(self.err.item())
        if e:
            raise _capi.DirHipError('dir_jpeg_decode_records: record %d of a batch did not describe a %dx%d image' % (e - 1, self.H, self.W))
