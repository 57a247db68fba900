// Winograd F(3x3, 2x2) for the 4x4 stride-2 pad-1 layers of G: the encoder's ConvBlock(k4, s2, p1, reflection) and the
// decoder's ConvTBlock = ConvTranspose2d(k4, s2, p1) (AE_model_unet.py:60-94, :497-520).  With the large windows in the
// frequency domain and the 3x3 layers on F(2x2,3x3) these eight layers were 26 % of the fp32 step on the direct kernels.
//
// A 4x4 stride-2 convolution is a 2x2 STRIDE-1 convolution over the four polyphase images of its (padded) input:
//     y[i][j] = sum_{a,b in {0,1}} sum_{u,v in {0,1}} X_ab[i+u][j+v] * w[2u+a][2v+b],   X_ab[m][n] = xpad[2m+a][2n+b]
// i.e. a 2x2 stride-1 layer with 4*Cin input channels, and F(3x3,2x2) computes a 3x3 output tile of such a layer from a
// 4x4 patch with 16 multiplies instead of 36 (the same interpolation points 0, +-1, inf as F(2x2,3x3): transforms that
// only add and halve, the same fp32 error class -- DESIGN.md 2.5/2.7).  16 MACs per output pixel and (cin, cout) pair
// become 7.1: a 2.25x cut, again executed by the per-bin fp32 MFMA GEMMs of wino_gemm.h.
//
//   form A  (strided conv forward; ConvTranspose data gradient)
//     input    x -> V [16][tile][4*Cx]     V = B^T X_ab B per phase, tile = 3x3 outputs = an 8x8 input footprint
//     weights  w -> U [16][Nout][4*Cx]     U = G g_ab G^T,  g_ab[u][v] = w[2u+a][2v+b]
//     gemm     Mo[bin] = V[bin] * U[bin]^T   (K = 4*Cx)
//     output   Mo -> y                       Y = A^T m A (3x3 per tile) + epilogue (BN partials, affine/ReLU, addsrc)
//   form B  (ConvTranspose forward; strided-conv data gradient): z[r] = sum_i d[i] w[r+1-2i].  Per output phase a' the taps
//     reduce to a 2-tap correlation of the zero-padded d (window dp[s..s+3] = d[s-1..s+2]):
//         z[2s+2p]   = (dp * (w3, w1))[s+p],   z[2s-1+2p] = (dp * (w2, w0))[s+p],   p = 0..2
//     so ONE transform of a 4x4 patch of d (tile stride 3) feeds all four output phases: a 6x6 output block at (6t-1, 6t-1).
//     input    d -> Vd [16][tile][Cd],  weights -> U' [16][4*Nout][Cd],  gemm (K = Cd, N = 4*Nout),  output -> 6x6 block
//     (cropped to the ConvTranspose output, or written to the (H+2)x(W+2) padded domain and folded for a reflection layer)
//   weight gradient: dg_ab = G^T [ sum_tiles (B^T X_ab B) (.) (A dy A^T) ] G  -- V of form A (kept from the forward for a
//     conv; the transform of dz, shared with the data gradient, for a ConvTranspose), dy lifted 3x3 -> 4x4, the reduction over
//     tiles on the MFMA (wino_gemm_tn_kernel), and a last small kernel that folds the 16 bins into the 2x2 taps of each phase.
//
// F(3,2):  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1],  G = [1 0; .5 .5; .5 -.5; 0 1],  A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1]
// (verified against the direct correlation in tests/test_wino2conv_model_cpu.py).
#include "common.h"
#include "wino_gemm.h"
#include "gemm_x3.h"

namespace {

struct W2Geom {
    int B;
    int Hx, Wx, Cx;               // the LARGE image (conv input / ConvTranspose output) and its channels
    int Hy, Wy, Cy;               // the SMALL image (conv output / ConvTranspose input): Hy = Hx / 2
    int reflect;                  // conv: reflection padding of x (pad 1); 0 = zeros
    int ta_y, ta_x, Ma;           // form A tiling: 3x3 tiles of the small image
    int tb_y, tb_x, Mb;           // form B tiling: 6x6 blocks of the large image (+ border when padded-domain)
    int xq, yq;                   // log2(Cx / 64), log2(Cy / 64)
    int x3;                       // per-bin GEMMs as bf16 x 3 split products unless GDN_HINT_NO_X3
};

// ---- form A input: V[bin][t][(a*2+b)*Cx + c] = (B^T X_ab B)[bin], X_ab[p][q] = xpad[6ty + 2p + a][6tx + 2q + b] ----
__global__ __launch_bounds__(256) void w2_input_a_kernel(const float* __restrict__ x, int ldx, float* __restrict__ V, W2Geom g) {
    int t, c;
    if (!wino_decode(g.Ma, g.xq, t, c)) return;
    const int tx = t % g.ta_x, ty = (t / g.ta_x) % g.ta_y, img = t / (g.ta_x * g.ta_y);
    const int lim = g.reflect ? 1 : 0;       // mirrored border row / column -1 and H (W); anything further out reads zero
    float P[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int iy = 6 * ty - 1 + i;
        const bool row_ok = iy >= -lim && iy < g.Hx + lim;
        const int iyr = iy < 0 ? -iy : (iy >= g.Hx ? 2 * g.Hx - 2 - iy : iy);
        const float* row = x + ((size_t)(img * g.Hx + (row_ok ? iyr : 0)) * g.Wx) * ldx + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ix = 6 * tx - 1 + j;
            const int ixr = ix < 0 ? -ix : (ix >= g.Wx ? 2 * g.Wx - 2 - ix : ix);
            P[i][j] = (row_ok && ix >= -lim && ix < g.Wx + lim) ? row[(size_t)ixr * ldx] : 0.f;
        }
    }
    const int K = 4 * g.Cx;
    const size_t bs = (size_t)g.Ma * K;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float r[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {          // B^T along rows of the phase image
                const float d0 = P[a][2 * q + b], d1 = P[2 + a][2 * q + b], d2 = P[4 + a][2 * q + b], d3 = P[6 + a][2 * q + b];
                r[0][q] = d0 - d2; r[1][q] = d1 + d2; r[2][q] = d2 - d1; r[3][q] = d3 - d1;
            }
            float* dst = V + (size_t)t * K + (a * 2 + b) * g.Cx + c;
#pragma unroll
            for (int p = 0; p < 4; ++p) {          // ... and along columns
                dst[0] = r[p][0] - r[p][2]; dst += bs; GDN_KEEP(dst);
                dst[0] = r[p][1] + r[p][2]; dst += bs; GDN_KEEP(dst);
                dst[0] = r[p][2] - r[p][1]; dst += bs; GDN_KEEP(dst);
                dst[0] = r[p][3] - r[p][1]; dst += bs; GDN_KEEP(dst);
            }
        }
}

// G g G^T of a 2x2 kernel -> 16 values written with stride bs
__device__ __forceinline__ void w2_store_u(float g00, float g01, float g10, float g11, float* dst, size_t bs) {
    const float r[4][2] = {{g00, g01}, {0.5f * (g00 + g10), 0.5f * (g01 + g11)}, {0.5f * (g00 - g10), 0.5f * (g01 - g11)}, {g10, g11}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[0] = r[i][0]; dst += bs; GDN_KEEP(dst);
        dst[0] = 0.5f * (r[i][0] + r[i][1]); dst += bs; GDN_KEEP(dst);
        dst[0] = 0.5f * (r[i][0] - r[i][1]); dst += bs; GDN_KEEP(dst);
        dst[0] = r[i][1]; dst += bs; GDN_KEEP(dst);
    }
}

// ---- form A weights: U[bin][n][(a*2+b)*Cx + c].  w is tap-major [16][R][S] (S contiguous); swap = 0: (n, c) = (R, S) index
// (strided conv forward), swap = 1: (n, c) = (S, R) (ConvTranspose data gradient: its weight is stored [tap][Cout_T][Cin_T]
// and the conv-form output channel is Cin_T) ----
__global__ __launch_bounds__(256) void w2_weights_a_kernel(const float* __restrict__ w, float* __restrict__ U, int Nout, int Cx, int swap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nout * Cx) return;
    // consecutive threads walk c, the contiguous index of the OUTPUT: the 64 stores coalesce (the 16 tap reads are strided
    // when swap, 4x fewer of them)
    const int c = i % Cx, n = i / Cx;
    const size_t ts = (size_t)Nout * Cx;
    const float* src = w + (swap ? (size_t)c * Nout + n : (size_t)n * Cx + c);
    float g[4][4];
#pragma unroll
    for (int ty = 0; ty < 4; ++ty)
#pragma unroll
        for (int tx = 0; tx < 4; ++tx) g[ty][tx] = src[(size_t)(ty * 4 + tx) * ts];
    const int K = 4 * Cx;
    const size_t bs = (size_t)Nout * K;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            w2_store_u(g[a][b], g[a][2 + b], g[2 + a][b], g[2 + a][2 + b], U + (size_t)n * K + (a * 2 + b) * Cx + c, bs);
}

// ---- form B weights: U'[bin][(a'*2+b')*Nout + n][cd],  g'_{a'b'}[u][v] = w[3 - a' - 2u][3 - b' - 2v].
// swap = 0: w[tap][R = n][S = cd] (ConvTranspose forward: [tap][Cout_T][Cin_T]);  swap = 1: w[tap][R = cd][S = n] (strided-conv
// data gradient: [tap][Cout][Cin], reduction over Cout) ----
__global__ __launch_bounds__(256) void w2_weights_b_kernel(const float* __restrict__ w, float* __restrict__ U, int Nout, int Cd, int swap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nout * Cd) return;
    const int cd = i % Cd, n = i / Cd;                       // cd: the contiguous index of the output
    const size_t ts = (size_t)Nout * Cd;
    const float* src = w + (swap ? (size_t)cd * Nout + n : (size_t)n * Cd + cd);
    float g[4][4];
#pragma unroll
    for (int ty = 0; ty < 4; ++ty)
#pragma unroll
        for (int tx = 0; tx < 4; ++tx) g[ty][tx] = src[(size_t)(ty * 4 + tx) * ts];
    const size_t bs = (size_t)4 * Nout * Cd;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            w2_store_u(g[3 - a][3 - b], g[3 - a][1 - b], g[1 - a][3 - b], g[1 - a][1 - b],
                       U + ((size_t)(a * 2 + b) * Nout + n) * Cd + cd, bs);
}

// shared epilogue of the two output kernels
struct W2Ep {
    const float* addsrc; int ld_add;
    float* stats;
    const float* ep_scale; const float* ep_shift;
    int act;
};

// ---- form A output: y[3ty + i][3tx + j][n] = (A^T m A)[i][j];  stats slot = group of 4 tiles ----
__global__ __launch_bounds__(256) void w2_output_a_kernel(const float* __restrict__ Mo, float* __restrict__ y, int ldy, W2Ep ep,
                                                          W2Geom g) {
    __shared__ float red[256 * 2];
    int t, n;
    const bool live = wino_decode(g.Ma, g.yq, t, n);
    float s1 = 0.f, s2 = 0.f;
    if (live) {
        const int tx = t % g.ta_x, ty = (t / g.ta_x) % g.ta_y, img = t / (g.ta_x * g.ta_y);
        float m[4][4];
        const float* src = Mo + (size_t)t * g.Cy + n;
        const size_t bs = (size_t)g.Ma * g.Cy;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { m[i][j] = *src; src += bs; GDN_KEEP(src); }
        float r[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[0][j] = m[0][j] + m[1][j] + m[2][j]; r[1][j] = m[1][j] - m[2][j]; r[2][j] = m[1][j] + m[2][j] + m[3][j]; }
        const float es = ep.ep_scale ? ep.ep_scale[n] : 1.f, et = ep.ep_shift ? ep.ep_shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float o[3] = {r[i][0] + r[i][1] + r[i][2], r[i][1] - r[i][2], r[i][1] + r[i][2] + r[i][3]};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int oy = 3 * ty + i, ox = 3 * tx + j;
                if (oy < g.Hy && ox < g.Wy) {
                    float val = o[j];
                    s1 += val; s2 += val * val;
                    if (ep.ep_scale) val = val * es + et;
                    if (ep.act & GDN_ACT_RELU) val = fmaxf(val, 0.f);
                    const size_t px = (size_t)(img * g.Hy + oy) * g.Wy + ox;
                    if (ep.addsrc) val += ep.addsrc[px * ep.ld_add + n];
                    y[px * ldy + n] = val;
                }
            }
        }
    }
    if (ep.stats) {
        red[threadIdx.x * 2] = s1; red[threadIdx.x * 2 + 1] = s2;
        __syncthreads();
        if (threadIdx.x < 64) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a1 += red[(j * 64 + threadIdx.x) * 2]; a2 += red[(j * 64 + threadIdx.x) * 2 + 1]; }
            const int slot = blockIdx.x >> g.yq;
            const int ch = (blockIdx.x & ((1 << g.yq) - 1)) * 64 + threadIdx.x;
            ep.stats[((size_t)slot * 2 + 0) * g.Cy + ch] = a1;
            ep.stats[((size_t)slot * 2 + 1) * g.Cy + ch] = a2;
        }
    }
}

// ---- form B input: Vd[bin][t][c] = (B^T D B)[bin], D[p][q] = d[3ty - 1 + p][3tx - 1 + q] (zero outside) ----
__global__ __launch_bounds__(256) void w2_input_b_kernel(const float* __restrict__ d, int ldd, float* __restrict__ Vd, W2Geom g) {
    int t, c;
    if (!wino_decode(g.Mb, g.yq, t, c)) return;
    const int tx = t % g.tb_x, ty = (t / g.tb_x) % g.tb_y, img = t / (g.tb_x * g.tb_y);
    float D[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int iy = 3 * ty - 1 + i;
        const bool row_ok = iy >= 0 && iy < g.Hy;
        const float* row = d + ((size_t)(img * g.Hy + (row_ok ? iy : 0)) * g.Wy) * ldd + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ix = 3 * tx - 1 + j;
            D[i][j] = (row_ok && ix >= 0 && ix < g.Wy) ? row[(size_t)ix * ldd] : 0.f;
        }
    }
    float r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[0][j] = D[0][j] - D[2][j]; r[1][j] = D[1][j] + D[2][j]; r[2][j] = D[2][j] - D[1][j]; r[3][j] = D[3][j] - D[1][j];
    }
    float* dst = Vd + (size_t)t * g.Cy + c;
    const size_t bs = (size_t)g.Mb * g.Cy;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[0] = r[i][0] - r[i][2]; dst += bs; GDN_KEEP(dst);
        dst[0] = r[i][1] + r[i][2]; dst += bs; GDN_KEEP(dst);
        dst[0] = r[i][2] - r[i][1]; dst += bs; GDN_KEEP(dst);
        dst[0] = r[i][3] - r[i][1]; dst += bs; GDN_KEEP(dst);
    }
}

// ---- form B output: the 6x6 block of tile t: out row 6ty + 2p - a' + off (off = 0: ConvTranspose output / zero-padded conv
// gradient, rows outside [0, Hout) dropped; off = 1: the (H+2) x (W+2) padded domain of a reflection layer) ----
__global__ __launch_bounds__(256) void w2_output_b_kernel(const float* __restrict__ Eo, float* __restrict__ out, int ldo, W2Ep ep,
                                                          W2Geom g, int Hout, int Wout, int off) {
    __shared__ float red[256 * 2];
    int t, n;
    const bool live = wino_decode(g.Mb, g.xq, t, n);
    float s1 = 0.f, s2 = 0.f;
    if (live) {
        const int tx = t % g.tb_x, ty = (t / g.tb_x) % g.tb_y, img = t / (g.tb_x * g.tb_y);
        const int N4 = 4 * g.Cx;
        const size_t bs = (size_t)g.Mb * N4;
        const float es = ep.ep_scale ? ep.ep_scale[n] : 1.f, et = ep.ep_shift ? ep.ep_shift[n] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float m[4][4];
                const float* src = Eo + (size_t)t * N4 + (a * 2 + b) * g.Cx + n;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { m[i][j] = *src; src += bs; GDN_KEEP(src); }
                float r[3][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { r[0][j] = m[0][j] + m[1][j] + m[2][j]; r[1][j] = m[1][j] - m[2][j]; r[2][j] = m[1][j] + m[2][j] + m[3][j]; }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float o[3] = {r[i][0] + r[i][1] + r[i][2], r[i][1] - r[i][2], r[i][1] + r[i][2] + r[i][3]};
                    const int oy = 6 * ty + 2 * i - a + off;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int ox = 6 * tx + 2 * j - b + off;
                        if (oy >= 0 && oy < Hout && ox >= 0 && ox < Wout) {
                            float val = o[j];
                            s1 += val; s2 += val * val;
                            if (ep.ep_scale) val = val * es + et;
                            if (ep.act & GDN_ACT_RELU) val = fmaxf(val, 0.f);
                            const size_t px = (size_t)(img * Hout + oy) * Wout + ox;
                            if (ep.addsrc) val += ep.addsrc[px * ep.ld_add + n];
                            out[px * ldo + n] = val;
                        }
                    }
                }
            }
    }
    if (ep.stats) {
        red[threadIdx.x * 2] = s1; red[threadIdx.x * 2 + 1] = s2;
        __syncthreads();
        if (threadIdx.x < 64) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a1 += red[(j * 64 + threadIdx.x) * 2]; a2 += red[(j * 64 + threadIdx.x) * 2 + 1]; }
            const int slot = blockIdx.x >> g.xq;
            const int ch = (blockIdx.x & ((1 << g.xq) - 1)) * 64 + threadIdx.x;
            ep.stats[((size_t)slot * 2 + 0) * g.Cx + ch] = a1;
            ep.stats[((size_t)slot * 2 + 1) * g.Cx + ch] = a2;
        }
    }
}

// ---- weight gradient, small-image operand: Dv[bin][t][n] = (A s A^T)[bin], s = the 3x3 tile of the small image (form A tiling) ----
__global__ __launch_bounds__(256) void w2_lift_kernel(const float* __restrict__ sm, int lds_, float* __restrict__ Dv, W2Geom g) {
    int t, n;
    if (!wino_decode(g.Ma, g.yq, t, n)) return;
    const int tx = t % g.ta_x, ty = (t / g.ta_x) % g.ta_y, img = t / (g.ta_x * g.ta_y);
    float y[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int oy = 3 * ty + i, ox = 3 * tx + j;
            y[i][j] = (oy < g.Hy && ox < g.Wy) ? sm[((size_t)(img * g.Hy + oy) * g.Wy + ox) * lds_ + n] : 0.f;
        }
    float s[4][3];                          // A along rows: (y0, y0 + y1 + y2, y0 - y1 + y2, y2)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        s[0][j] = y[0][j]; s[1][j] = y[0][j] + y[1][j] + y[2][j]; s[2][j] = y[0][j] - y[1][j] + y[2][j]; s[3][j] = y[2][j];
    }
    float* dst = Dv + (size_t)t * g.Cy + n;
    const size_t bs = (size_t)g.Ma * g.Cy;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[0] = s[i][0]; dst += bs; GDN_KEEP(dst);
        dst[0] = s[i][0] + s[i][1] + s[i][2]; dst += bs; GDN_KEEP(dst);
        dst[0] = s[i][0] - s[i][1] + s[i][2]; dst += bs; GDN_KEEP(dst);
        dst[0] = s[i][2]; dst += bs; GDN_KEEP(dst);
    }
}

// ---- weight gradient output: dw[(2u+a)*4 + (2v+b)][..] = (G^T P_ab G)[u][v],  P[bin][n][(a*2+b)*Cx + c].
// transposed = 0: dw[tap][n][c] (conv);  1: dw[tap][c][n] (ConvTranspose: [tap][Cout_T = large-image channel][Cin_T]) ----
// grid.y = the phase (a, b): the smallest layer (64 x 128 channel pairs) is 32 workgroups per phase, and its reduction over up to
// 16 splits is 16 independent loads per split -- issued together, summed in split order (round 3: was one thread per channel
// pair walking all four phases with one dependent load at a time, 235 us for 0.5 MB of output).
__global__ __launch_bounds__(256) void w2_wgrad_output_kernel(const float* __restrict__ P, float* __restrict__ dw, int Ny, int Cx, int transposed,
                                                              int nsplit) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Ny * Cx) return;
    const int c = i % Cx, n = i / Cx;
    const int a = blockIdx.y >> 1, b = blockIdx.y & 1;
    const int K = 4 * Cx;
    const size_t bs = (size_t)Ny * K, ts = (size_t)Ny * Cx;
    const size_t o = transposed ? (size_t)c * Ny + n : (size_t)n * Cx + c;
    float p[16];
    const float* src = P + (size_t)n * K + (a * 2 + b) * Cx + c;
#pragma unroll
    for (int q = 0; q < 16; ++q) p[q] = src[(size_t)q * bs];
    for (int sp = 1; sp < nsplit; ++sp) {                       // split reduction, fixed order
        src += (size_t)WINO_BINS * bs;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = src[(size_t)q * bs];
#pragma unroll
        for (int q = 0; q < 16; ++q) p[q] += v[q];
    }
    float r[2][4];                  // G^T along rows: (p0 + (p1 + p2)/2, (p1 - p2)/2 + p3)
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[0][e] = p[e] + 0.5f * (p[4 + e] + p[8 + e]); r[1][e] = 0.5f * (p[4 + e] - p[8 + e]) + p[12 + e]; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        dw[(size_t)((2 * u + a) * 4 + b) * ts + o] = r[u][0] + 0.5f * (r[u][1] + r[u][2]);
        dw[(size_t)((2 * u + a) * 4 + 2 + b) * ts + o] = 0.5f * (r[u][1] - r[u][2]) + r[u][3];
    }
}

bool w2_geom(const gdn_conv_geom* g, W2Geom& f) {
    if (!g || g->k != 4 || g->stride != 2 || g->pad != 1) return false;
    if (g->transposed && g->pad_mode != 0) return false;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if ((g->Cin % 64) || (g->Cout % 64) || g->Cin > 512 || g->Cout > 512 || !pow2(g->Cin / 64) || !pow2(g->Cout / 64)) return false;
    f.B = g->B;
    if (g->transposed) {            // ConvTranspose2d: input (small) [B,H,W,Cin] -> output (large) [B,2H,2W,Cout]
        f.Hy = g->H; f.Wy = g->W; f.Cy = g->Cin; f.Hx = 2 * g->H; f.Wx = 2 * g->W; f.Cx = g->Cout;
    } else {                        // Conv2d: input (large) [B,H,W,Cin] -> output (small) [B,H/2,W/2,Cout]
        if ((g->H & 1) || (g->W & 1) || g->H < 4 || g->W < 4) return false;
        f.Hx = g->H; f.Wx = g->W; f.Cx = g->Cin; f.Hy = g->H / 2; f.Wy = g->W / 2; f.Cy = g->Cout;
    }
    f.reflect = (!g->transposed && g->pad_mode == 1) ? 1 : 0;
    f.ta_y = cdiv(f.Hy, 3); f.ta_x = cdiv(f.Wy, 3); f.Ma = f.B * f.ta_y * f.ta_x;
    // form B blocks start at row -1: rows -1 .. Hx (padded-domain gradient of a reflection layer) or 0 .. Hx-1
    const int ext = f.reflect ? 2 : 1;
    f.tb_y = cdiv(f.Hx + ext, 6); f.tb_x = cdiv(f.Wx + ext, 6); f.Mb = f.B * f.tb_y * f.tb_x;
    f.xq = 0; while ((64 << f.xq) < f.Cx) ++f.xq;
    f.yq = 0; while ((64 << f.yq) < f.Cy) ++f.yq;
    f.x3 = (g->hints & GDN_HINT_NO_X3) ? 0 : 1;
    return true;
}

inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }
inline size_t va_bytes(const W2Geom& f) { return al256((size_t)WINO_BINS * f.Ma * 4 * f.Cx * 4); }      // V, form A
inline size_t ua_bytes(const W2Geom& f) { return al256((size_t)WINO_BINS * f.Cy * 4 * f.Cx * 4); }      // U / U' / P (same size)
inline size_t ma_bytes(const W2Geom& f) { return al256((size_t)WINO_BINS * f.Ma * f.Cy * 4); }          // Mo / Dv, form A
inline size_t vb_bytes(const W2Geom& f) { return al256((size_t)WINO_BINS * f.Mb * f.Cy * 4); }          // Vd, form B
inline size_t eb_bytes(const W2Geom& f) { return al256((size_t)WINO_BINS * f.Mb * 4 * f.Cx * 4); }      // Eo, form B
// the U region also holds the bf16 x 3 panels of the transformed weights (gemm_x3.h; packed from the fp32 set by one small
// launch): rows = Cy, k = 4 Cx (form A) or rows = 4 Cx, k = Cy (form B), behind the fp32 set
inline size_t ux_bytes(const W2Geom& f) {
    const size_t pa = (size_t)WINO_BINS * x3_packed_bytes(f.Cy, 4 * f.Cx), pb = (size_t)WINO_BINS * x3_packed_bytes(4 * f.Cx, f.Cy);
    return ua_bytes(f) + al256(pa > pb ? pa : pb);
}
inline bool w2_tn_x3(const W2Geom& f) { return f.x3 && gemm_x3_tn_ok(f.Ma, f.Cy, 4 * f.Cx); }
inline int w2_splits(const W2Geom& f) {
    return w2_tn_x3(f) ? gemm_x3_tn_splits(WINO_BINS, f.Ma, f.Cy, 4 * f.Cx) : wino_tn_splits(f.Ma, f.Cy, 4 * f.Cx);
}
inline int w2_splits_max(const W2Geom& f) {
    const int a = wino_tn_splits(f.Ma, f.Cy, 4 * f.Cx);
    const int b = gemm_x3_tn_ok(f.Ma, f.Cy, 4 * f.Cx) ? gemm_x3_tn_splits(WINO_BINS, f.Ma, f.Cy, 4 * f.Cx) : 1;
    return a > b ? a : b;
}
void w2_gemm_tn(const W2Geom& f, const float* Dv, const float* V, float* P, int ns, hipStream_t st) {
    if (w2_tn_x3(f)) launch_gemm_x3_tn(Dv, V, P, WINO_BINS, f.Ma, f.Cy, 4 * f.Cx, ns, st);
    else launch_wino_gemm_tn(Dv, V, P, f.Ma, f.Cy, 4 * f.Cx, ns, st);
}
inline size_t p_bytes(const W2Geom& f) {                                                                // P, one set per split
    const size_t pr = (size_t)w2_splits_max(f) * ua_bytes(f);
    return pr > ux_bytes(f) ? pr : ux_bytes(f);
}
// the 16 per-bin GEMMs Cm = A * U^T: as bf16 x 3 split products on the bf16 matrix pipe where the shape allows, else fp32 MFMA
void w2_gemm(const W2Geom& f, const float* A, float* U, float* Cm, int M, int N, int K, hipStream_t st) {
    if (f.x3 && gemm_x3_ok(M, N, K)) {
        void* Up = (char*)U + ua_bytes(f);
        launch_x3_pack_rows(U, Up, WINO_BINS, N, K, st);
        launch_gemm_x3_nt(A, Up, Cm, WINO_BINS, M, N, K, st);
    } else {
        launch_wino_gemm(A, (const float*)U, Cm, M, N, K, st);
    }
}
inline size_t pad_bytes(const W2Geom& f) { return f.reflect ? al256((size_t)f.B * (f.Hx + 2) * (f.Wx + 2) * f.Cx * 4) : 0; }

void run_form_a(const W2Geom& f, const float* x, int ldx, const float* w, int swap, float* V, float* U, float* Mo, float* y, int ldy,
                const W2Ep& ep, hipStream_t st) {
    hipLaunchKernelGGL(w2_input_a_kernel, dim3(cdiv(f.Ma, 4) << f.xq), dim3(256), 0, st, x, ldx, V, f);
    hipLaunchKernelGGL(w2_weights_a_kernel, dim3(cdiv(f.Cy * f.Cx, 256)), dim3(256), 0, st, w, U, f.Cy, f.Cx, swap);
    w2_gemm(f, (const float*)V, U, Mo, f.Ma, f.Cy, 4 * f.Cx, st);
    hipLaunchKernelGGL(w2_output_a_kernel, dim3(cdiv(f.Ma, 4) << f.yq), dim3(256), 0, st, (const float*)Mo, y, ldy, ep, f);
}

void run_form_b(const W2Geom& f, const float* d, int ldd, const float* w, int swap, float* Vd, float* U, float* Eo, float* out, int ldo,
                int Hout, int Wout, int off, const W2Ep& ep, hipStream_t st) {
    hipLaunchKernelGGL(w2_input_b_kernel, dim3(cdiv(f.Mb, 4) << f.yq), dim3(256), 0, st, d, ldd, Vd, f);
    hipLaunchKernelGGL(w2_weights_b_kernel, dim3(cdiv(f.Cx * f.Cy, 256)), dim3(256), 0, st, w, U, f.Cx, f.Cy, swap);
    w2_gemm(f, (const float*)Vd, U, Eo, f.Mb, 4 * f.Cx, f.Cy, st);
    hipLaunchKernelGGL(w2_output_b_kernel, dim3(cdiv(f.Mb, 4) << f.xq), dim3(256), 0, st, (const float*)Eo, out, ldo, ep, f, Hout, Wout, off);
}

}  // namespace

// saved state of a Conv2d forward for its weight gradient: V (the transformed input).  0 for a ConvTranspose2d, whose weight
// gradient transforms the output gradient instead.
extern "C" size_t gdn_wino2conv_state_bytes(const gdn_conv_geom* g) {
    W2Geom f;
    if (!w2_geom(g, f)) return 0;
    return g->transposed ? 0 : va_bytes(f);
}

extern "C" size_t gdn_wino2conv_fwd_workspace_bytes(const gdn_conv_geom* g) {
    W2Geom f;
    if (!w2_geom(g, f)) return 0;
    return g->transposed ? vb_bytes(f) + ux_bytes(f) + eb_bytes(f) : va_bytes(f) + ux_bytes(f) + ma_bytes(f);
}

extern "C" int64_t gdn_wino2conv_stats_slots(const gdn_conv_geom* g) {
    W2Geom f;
    if (!w2_geom(g, f)) return GDN_ERR_UNSUPPORTED;
    return cdiv(g->transposed ? f.Mb : f.Ma, 4);
}

extern "C" int gdn_wino2conv_fwd(const gdn_conv_geom* g, const float* x, int32_t ldx, const float* w, float* y, int32_t ldy,
                                 const float* addsrc, int32_t ld_add, float* stats, const float* ep_scale,
                                 const float* ep_shift, int32_t act, void* state_out, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    W2Geom f;
    if (!w2_geom(g, f)) return GDN_ERR_UNSUPPORTED;
    if (!x || !w || !y || (!ep_scale) != (!ep_shift) || (act & GDN_ACT_TANH)) return GDN_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gdn_wino2conv_fwd_workspace_bytes(g)) return GDN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const W2Ep ep = {addsrc, ld_add, stats, ep_scale, ep_shift, act};
    char* p = (char*)workspace;
    if (!g->transposed) {
        float* V = (float*)p; p += va_bytes(f);
        float* U = (float*)p; p += ux_bytes(f);
        float* Mo = (float*)p;
        if (state_out) V = (float*)state_out;
        run_form_a(f, x, ldx, w, 0, V, U, Mo, y, ldy, ep, st);
    } else {
        float* Vd = (float*)p; p += vb_bytes(f);
        float* U = (float*)p; p += ux_bytes(f);
        float* Eo = (float*)p;
        run_form_b(f, x, ldx, w, 0, Vd, U, Eo, y, ldy, f.Hx, f.Wx, 0, ep, st);
    }
    return gdn_launch_status();
}

// workspace: [V / Vd] [U or P] [Mo / Eo / Dv] (+ padded-domain gradient of a reflection layer)
extern "C" size_t gdn_wino2conv_bwd_workspace_bytes(const gdn_conv_geom* g) {
    W2Geom f;
    if (!w2_geom(g, f)) return 0;
    if (g->transposed) return va_bytes(f) + p_bytes(f) + ma_bytes(f);           // form A on dz; Dv reuses the Mo region
    const size_t a = vb_bytes(f) > ma_bytes(f) ? vb_bytes(f) : ma_bytes(f);     // Vd (data gradient) / Dv (weight gradient)
    return a + p_bytes(f) + eb_bytes(f) + pad_bytes(f);
}

// Conv2d (transposed = 0):       dy [B,H/2,W/2,Cout]; dx [B,H,W,Cin] = dgrad (+ addsrc) when dx != NULL (needs w);
//                                dw[tap][Cout][Cin] when dw != NULL (needs state = V of the forward)
// ConvTranspose2d (transposed=1): dy [B,2H,2W,Cout]; dx [B,H,W,Cin] (needs w); dw[tap][Cout][Cin] needs x, the forward input
extern "C" int gdn_wino2conv_bwd(const gdn_conv_geom* g, const float* dy, int32_t ldy, const float* w, const float* x,
                                 int32_t ldx_in, const void* state, float* dx, int32_t ldx, const float* addsrc,
                                 int32_t ld_add, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    W2Geom f;
    if (!w2_geom(g, f)) return GDN_ERR_UNSUPPORTED;
    if (!dy || (!dx && !dw) || (dx && !w)) return GDN_ERR_BAD_ARG;
    if (dw && (g->transposed ? !x : !state)) return GDN_ERR_BAD_ARG;
    if (dx && f.reflect && ((ldx % 4) || (addsrc && (ld_add % 4)))) return GDN_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < gdn_wino2conv_bwd_workspace_bytes(g)) return GDN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)workspace;
    if (g->transposed) {
        // dz = dy is the LARGE image: one form-A transform of it feeds the data gradient (a strided conv of dz with the
        // role-swapped weights) and the weight gradient (reduction against the lifted 3x3 tiles of the forward input x)
        float* V = (float*)p; p += va_bytes(f);
        float* U = (float*)p; p += p_bytes(f);
        float* Mo = (float*)p;
        hipLaunchKernelGGL(w2_input_a_kernel, dim3(cdiv(f.Ma, 4) << f.xq), dim3(256), 0, st, dy, ldy, V, f);
        if (dx) {
            hipLaunchKernelGGL(w2_weights_a_kernel, dim3(cdiv(f.Cy * f.Cx, 256)), dim3(256), 0, st, w, U, f.Cy, f.Cx, 1);
            w2_gemm(f, (const float*)V, U, Mo, f.Ma, f.Cy, 4 * f.Cx, st);
            const W2Ep ep = {addsrc, ld_add, nullptr, nullptr, nullptr, 0};
            hipLaunchKernelGGL(w2_output_a_kernel, dim3(cdiv(f.Ma, 4) << f.yq), dim3(256), 0, st, (const float*)Mo, dx, ldx, ep, f);
        }
        if (dw) {
            float* Dv = Mo;             // (stream-ordered after the output transform above)
            float* P = U;
            hipLaunchKernelGGL(w2_lift_kernel, dim3(cdiv(f.Ma, 4) << f.yq), dim3(256), 0, st, x, ldx_in, Dv, f);
            const int ns = w2_splits(f);
            w2_gemm_tn(f, (const float*)Dv, (const float*)V, P, ns, st);
            hipLaunchKernelGGL(w2_wgrad_output_kernel, dim3(cdiv(f.Cy * f.Cx, 256), 4), dim3(256), 0, st, (const float*)P, dw, f.Cy, f.Cx, 1, ns);
        }
        return gdn_launch_status();
    }
    const size_t a = vb_bytes(f) > ma_bytes(f) ? vb_bytes(f) : ma_bytes(f);
    float* Vd = (float*)p; p += a;
    float* U = (float*)p; p += p_bytes(f);
    float* Eo = (float*)p; p += eb_bytes(f);
    float* dxp = (float*)p;
    if (dw) {
        float* Dv = Vd;
        float* P = U;
        hipLaunchKernelGGL(w2_lift_kernel, dim3(cdiv(f.Ma, 4) << f.yq), dim3(256), 0, st, dy, ldy, Dv, f);
        const int ns = w2_splits(f);
        w2_gemm_tn(f, (const float*)Dv, (const float*)state, P, ns, st);
        hipLaunchKernelGGL(w2_wgrad_output_kernel, dim3(cdiv(f.Cy * f.Cx, 256), 4), dim3(256), 0, st, (const float*)P, dw, f.Cy, f.Cx, 0, ns);
    }
    if (dx) {
        if (f.reflect) {
            // gradient over the (H+2) x (W+2) padded domain, then the border rows / columns folded onto the rows they mirror
            const W2Ep ep = {nullptr, 0, nullptr, nullptr, nullptr, 0};
            run_form_b(f, dy, ldy, w, 1, Vd, U, Eo, dxp, f.Cx, f.Hx + 2, f.Wx + 2, 1, ep, st);
            const int64_t nb = cdiv64((int64_t)f.B * f.Hx * f.Wx * (f.Cx / 4), 256);
            hipLaunchKernelGGL(wino_reflect_fold_kernel, dim3((unsigned)(nb < 65536 * 8 ? nb : 65536 * 8)), dim3(256), 0, st,
                               (const float*)dxp, dx, ldx, addsrc, ld_add, f.B, f.Hx, f.Wx, f.Cx);
        } else {
            const W2Ep ep = {addsrc, ld_add, nullptr, nullptr, nullptr, 0};
            run_form_b(f, dy, ldy, w, 1, Vd, U, Eo, dx, ldx, f.Hx, f.Wx, 0, ep, st);
        }
    }
    return gdn_launch_status();
}
