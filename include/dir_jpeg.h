/* libdir_jpeg.so -- the HOST half of the from-files input path: entropy (Huffman) decode of a baseline JPEG into a coefficient record.
 * Plain C ABI, no GPU runtime (the decode worker processes load it); the GPU half is dir_jpeg_decode_records in dir_hip.h.
 *
 * Replaces, together with dir_jpeg_decode_records, the reference's  cv.imread(<split>/img/<idx>.jpg)  (apps/eval.py:56, dataset/interhand.py:223)
 * over the files of dataset/prepare_data.py:123-166.  The reference-side binding a maintainer would add is in INTEGRATION.md. */
#ifndef DIR_JPEG_H_
#define DIR_JPEG_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIR_JPEG_ABI_VERSION 1
#define DIR_JPEG_MAGIC 0x4a524944 /* "DIRJ" */
#define DIR_JPEG_MAGIC_PIXELS 0x50524944 /* "DIRP": the record carries DECODED pixels instead (below) */
#define DIR_JPEG_OK 0
#define DIR_JPEG_E_ARG (-1)         /* null pointer / record smaller than a header                                    */
#define DIR_JPEG_E_FORMAT (-2)      /* not a JPEG, truncated or corrupt                                               */
#define DIR_JPEG_E_UNSUPPORTED (-3) /* progressive / arithmetic / lossless / multi-scan / 12-bit / unusual sampling   */
#define DIR_JPEG_E_SPACE (-4)       /* the record buffer is too small for this image's coefficients                   */

/* One image = this 512-byte header followed by int16 coefficients [component][block row][block column][64], QUANTISED (as coded), in natural
 * (row-major 8x8) order.  Components are Y, Cb, Cr (or Y alone); chroma sampling factors are 1x1, luma 1x1 (4:4:4), 2x1 (4:2:2) or 2x2 (4:2:0). */
typedef struct dir_jpeg_header {
    int32_t magic, width, height, ncomp;
    int32_t hmax, vmax, mcux, mcuy;           /* largest sampling factors; MCUs per row / column                               */
    int32_t h[3], v[3];                       /* sampling factors per component                                               */
    int32_t blocks_x[3], blocks_y[3];         /* blocks per row / column of each component (whole MCUs: includes the padding)  */
    int32_t coef_offset[3];                   /* first coefficient of the component, in int16 units after this header          */
    int32_t total_coef;                       /* int16 values that follow the header                                           */
    uint16_t quant[3][64];                    /* the component's quantisation table, natural order                             */
    int32_t reserved[8];
} dir_jpeg_header;                            /* sizeof == 512 */

/* A record whose magic is DIR_JPEG_MAGIC_PIXELS carries, after the same 512-byte header (width, height set; everything else zero), the decoded
 * frame itself as uint8 BGR [height][width][3]: what the host falls back to for files this decoder refuses (progressive, CMYK, ...) or that need
 * a resize -- dir_jpeg_decode_records copies such a frame through.  A 4:2:0 record of a W x H image has exactly the size of that frame + header. */

/* bytes of one JPEG -> record (header + coefficients).  Returns DIR_JPEG_OK or a negative DIR_JPEG_E_*.  Thread-safe, no global state. */
int dir_jpeg_decode_coefficients(const uint8_t* data, size_t n, void* record, size_t record_bytes);
/* record size for an image of this geometry (hsamp / vsamp: the luma sampling factors, 1 or 2; ncomp 1 or 3); 0 for bad arguments */
size_t dir_jpeg_record_bytes(int width, int height, int hsamp, int vsamp, int ncomp);
int dir_jpeg_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
