// Backward of RegressorOffset's three Linears (models/dir.py:339-351): given the gradients w.r.t. pd_mano_para_{left,right} [B,64]
// (from dir_mano_backward_pair) and pd_offset [B,3] (from dir_stage_losses_backward), the parameter gradients in the PARAMETERS'
// own layout (nn.Linear weight [out][in], bias [out]: they can be written straight into dir_amd.optim.FlatAdamW's gradient views)
// and the gradient w.r.t. the joint tokens [B,42,64] (the STE head's output).  The previous stage's mano_para / offset enter the
// Linears detached (models/dir.py:344-345,447-453): no gradient leaves through them.
//   global_feat_h = cat(tok_h.reshape(B, 1344), prev_para_h)            -> para_h = W_h global_feat_h + b_h        (h = left, right)
//   global_feat   = cat(tok_l.reshape(B, 1344), tok_r.reshape(..), prev_offset) -> offset = W_o global_feat + b_o
// Two launches: parameter gradients (one thread per weight element, fixed-order sum over the batch: deterministic) and token
// gradients (one thread per token feature).  HBM-trivial (0.7 MB of parameters, B x 11 KB of activations).
#include "dir_common.h"

namespace {

constexpr int TOKH = 21 * 64;             // 1344 token features per hand
constexpr int KH = TOKH + 64;             // 1408 inputs of mano_left / mano_right
constexpr int KO = 2 * TOKH + 3;          // 2691 inputs of offset

struct RegBwdArgs {
    const float* tok; const float* prev_para[2]; const float* prev_off;
    const float* g_para[2]; const float* g_off;
    const float* w[2]; const float* w_off;
    float* gw[2]; float* gb[2]; float* gw_off; float* gb_off; float* g_tok;
    int B;
};

// rows 0..63 mano_left, 64..127 mano_right, 128..130 offset; column k < K: weight element, k == K: bias
__global__ __launch_bounds__(256) void regress_bwd_param_kernel(RegBwdArgs a) {
    const int row = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    const int grp = row < 64 ? 0 : (row < 128 ? 1 : 2), o = grp == 2 ? row - 128 : (row & 63);
    const int K = grp == 2 ? KO : KH;
    if (k > K) return;
    const float* g = grp == 2 ? a.g_off + o : a.g_para[grp] + o;
    const int gs = grp == 2 ? 3 : 64;
    float acc = 0.f;
    for (int b = 0; b < a.B; ++b) {
        float x;
        if (k == K) x = 1.f;
        else if (grp < 2) x = k < TOKH ? a.tok[((size_t)b * 2 + grp) * TOKH + k] : a.prev_para[grp][(size_t)b * 64 + k - TOKH];
        else x = k < 2 * TOKH ? a.tok[(size_t)b * 2 * TOKH + k] : a.prev_off[(size_t)b * 3 + k - 2 * TOKH];
        acc = fmaf(g[(size_t)b * gs], x, acc);
    }
    if (k == K) (grp == 2 ? a.gb_off : a.gb[grp])[o] = acc;
    else (grp == 2 ? a.gw_off : a.gw[grp])[(size_t)o * K + k] = acc;
}

// g tok[b][hand][k] = sum_o g_para_hand[b][o] W_hand[o][k] + sum_c g_off[b][c] W_off[c][hand * 1344 + k]
__global__ __launch_bounds__(256) void regress_bwd_input_kernel(RegBwdArgs a) {
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * TOKH) return;
    const int hand = t >= TOKH, k = t - hand * TOKH;
    const float* g = a.g_para[hand] + (size_t)b * 64;
    const float* w = a.w[hand] + k;
    float acc = 0.f;
#pragma unroll 8
    for (int o = 0; o < 64; ++o) acc = fmaf(g[o], w[(size_t)o * KH], acc);
#pragma unroll
    for (int c = 0; c < 3; ++c) acc = fmaf(a.g_off[(size_t)b * 3 + c], a.w_off[(size_t)c * KO + t], acc);
    a.g_tok[(size_t)b * 2 * TOKH + t] = acc;
}

}  // namespace

extern "C" int dir_regress_backward(const float* w_left, const float* w_right, const float* w_offset, const float* tok,
                                    const float* prev_para_left, const float* prev_para_right, const float* prev_offset,
                                    const float* g_para_left, const float* g_para_right, const float* g_offset,
                                    float* gw_left, float* gb_left, float* gw_right, float* gb_right, float* gw_offset, float* gb_offset,
                                    float* g_tok, int B, void* stream) {
    using namespace dir;
    DIR_REQUIRE(w_left && w_right && w_offset && tok && prev_para_left && prev_para_right && prev_offset && g_para_left && g_para_right && g_offset,
                "dir_regress_backward: null input");
    DIR_REQUIRE(B > 0, "dir_regress_backward: bad B");
    RegBwdArgs a;
    a.tok = tok; a.prev_para[0] = prev_para_left; a.prev_para[1] = prev_para_right; a.prev_off = prev_offset;
    a.g_para[0] = g_para_left; a.g_para[1] = g_para_right; a.g_off = g_offset;
    a.w[0] = w_left; a.w[1] = w_right; a.w_off = w_offset;
    a.gw[0] = gw_left; a.gw[1] = gw_right; a.gb[0] = gb_left; a.gb[1] = gb_right; a.gw_off = gw_offset; a.gb_off = gb_offset; a.g_tok = g_tok;
    a.B = B;
    hipStream_t s = (hipStream_t)stream;
    if (gw_left || gw_right || gw_offset) {
        DIR_REQUIRE(gw_left && gw_right && gw_offset && gb_left && gb_right && gb_offset, "dir_regress_backward: parameter gradients go together");
        DIR_LAUNCH(regress_bwd_param_kernel, dim3((KO + 1 + 255) / 256, 131), dim3(256), 0, s, a);
    }
    if (g_tok) DIR_LAUNCH(regress_bwd_input_kernel, dim3((2 * TOKH + 255) / 256, B), dim3(256), 0, s, a);
    return check_launch("dir_regress_backward");
}
