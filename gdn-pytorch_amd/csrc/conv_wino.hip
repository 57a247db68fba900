// Winograd F(2x2, 3x3) convolution for the 512-channel 3x3 residual layers (levels 3 and 4 of G and R: after the
// frequency-domain path took the large windows they are the biggest share of the fp32 step).
//
// A 3x3 stride-1 convolution costs 9 MACs per (pixel, cin, cout); F(2x2,3x3) costs 16 per 2x2 outputs = 4, a 2.25x
// cut with transforms that only add and halve -- the fp32 error stays at the level of the direct sum (rms 1.6x,
// DESIGN.md 2.5), unlike F(4x4,3x3).  The frequency-domain path of conv_fft.hip does not apply here: with 512 channels
// its weight spectrum would be 1.7 GB per layer.
//
//   input     x  -> V [16][tile][Cin]     V = B^T d B of every 4x4 patch (stride 2, zero border)
//   weights   w  -> U [16][Cout][Cin]     U = G g G^T   (data gradient: taps flipped, channel roles swapped)
//   gemm      Mo[bin] = V[bin] * U[bin]^T           16 real GEMMs, M = tiles, K = Cin, N = Cout (fp32 MFMA)
//   output    Mo -> y                                y = A^T m A (2x2 per tile) + the conv_igemm epilogue (BN partials,
//                                                    affine / ReLU / residual)
// Backward: the data gradient is the same pipeline on dy with the flipped / transposed weights; the weight gradient is
// dW = G^T [ sum_tiles (B^T d B) (.) (A dy A^T) ] G: V is kept from the forward, dy gets the 2x2 -> 4x4 transform, the sum
// over tiles is a reduction GEMM per bin on the MFMA, and a last small kernel folds the 16 bins into the 9 taps.
#include "common.h"
#include "wino_gemm.h"
#include "up2x.h"
#include "gemm_x3.h"

namespace {

struct WinoGeom {
    int B, H, W, C, N;            // input [B,H,W,C] -> output [B,Ho,Wo,N]
    int Ho, Wo;                   // output extent (= H, W except for the padded-domain data gradient of a reflection layer)
    int pad_off;                  // output (oy, ox) reads input rows oy - pad_off .. oy - pad_off + 2
    int reflect;                  // input border: 0 zeros, 1 mirror (ReflectionPad2d(1) + conv)
    int tiles_y, tiles_x, M;      // 2x2 output tiles; M = B * tiles_y * tiles_x
    int cq_shift, nq_shift;       // log2(C / 64), log2(N / 64)
    int x3;                       // per-bin GEMMs as bf16 x 3 split products (gemm_x3.h) unless the caller set GDN_HINT_NO_X3
    int T, bins;                  // outputs per tile side and transform bins: F(2x2,3x3) T = 2, 16 bins; F(4x4,3x3) T = 4, 36 bins
};

// V = B^T d B of the 4x4 patch whose top-left corner is (2a - pad_off, 2b - pad_off)
// in_scale != NULL: x is the RAW output of the producer convolution and the layer input is
// [relu](x * in_scale[c] + in_shift[c]) -- the producer's train-mode BatchNorm (+ReLU) applied on load, so that
// activation is never written to memory (ResidualBlock AE_model_unet.py:49-54).  Padding stays zero.
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int ldx, float* __restrict__ V, WinoGeom g,
                                                         const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                         int in_relu, int up2x) {
    int t, c;
    if (!wino_decode(g.M, g.cq_shift, t, c)) return;
    const float is = in_scale ? in_scale[c] : 1.f, it = in_scale ? in_shift[c] : 0.f;
    const float lo = in_relu ? 0.f : -3.402823466e38f;
    const int b2 = t % g.tiles_x, a2 = (t / g.tiles_x) % g.tiles_y, img = t / (g.tiles_x * g.tiles_y);
    float d[4][4];
    const int lim = g.reflect ? 1 : 0;      // mirrored border rows / columns -1 and H (W); anything further out reads zero
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int iy = 2 * a2 - g.pad_off + i;
        const bool row_ok = iy >= -lim && iy < g.H + lim;
        const int iyr = iy < 0 ? -iy : (iy >= g.H ? 2 * g.H - 2 - iy : iy);
        if (up2x) {
            // x is the LOW-resolution tensor [B][H/2][W/2][ldx]: the x2 bilinear upsampling happens here (up2x.h)
            const int Hl = g.H >> 1, Wl = g.W >> 1;
            float ly; int y0, y1;
            up_src(row_ok ? iyr : 0, Hl, up2x - 1, ly, y0, y1);
            const float* r0 = x + ((size_t)(img * Hl + y0) * Wl) * ldx + c;
            const float* r1 = x + ((size_t)(img * Hl + y1) * Wl) * ldx + c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ix = 2 * b2 - g.pad_off + j;
                const int ixr = ix < 0 ? -ix : (ix >= g.W ? 2 * g.W - 2 - ix : ix);
                const bool ok = row_ok && ix >= -lim && ix < g.W + lim;
                float v = 0.f;
                if (ok) {
                    float lx; int x0, x1;
                    up_src(ixr, Wl, up2x - 1, lx, x0, x1);
                    v = up2x_at(r0, r1, ldx, ly, x0, x1, lx);
                }
                d[i][j] = v;
            }
            continue;
        }
        const float* row = x + ((size_t)(img * g.H + (row_ok ? iyr : 0)) * g.W) * ldx + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ix = 2 * b2 - g.pad_off + j;
            const int ixr = ix < 0 ? -ix : (ix >= g.W ? 2 * g.W - 2 - ix : ix);
            const bool ok = row_ok && ix >= -lim && ix < g.W + lim;
            float v = ok ? row[(size_t)ixr * ldx] : 0.f;
            if (in_scale) v = ok ? fmaxf(v * is + it, lo) : 0.f;
            d[i][j] = v;
        }
    }
    float r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // B^T along rows
        r[0][j] = d[0][j] - d[2][j];
        r[1][j] = d[1][j] + d[2][j];
        r[2][j] = d[2][j] - d[1][j];
        r[3][j] = d[1][j] - d[3][j];
    }
    float* dst = V + (size_t)t * g.C + c;
    const size_t bs = (size_t)g.M * g.C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // ... and along columns
        dst[0] = r[i][0] - r[i][2]; dst += bs; GDN_KEEP(dst);
        dst[0] = r[i][1] + r[i][2]; dst += bs; GDN_KEEP(dst);
        dst[0] = r[i][2] - r[i][1]; dst += bs; GDN_KEEP(dst);
        dst[0] = r[i][1] - r[i][3]; dst += bs; GDN_KEEP(dst);
    }
}

// Dv = A dy A^T of the 2x2 output tile (weight gradient)
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, int ldy, float* __restrict__ Dv, WinoGeom g) {
    int t, n;
    if (!wino_decode(g.M, g.nq_shift, t, n)) return;
    const int b2 = t % g.tiles_x, a2 = (t / g.tiles_x) % g.tiles_y, img = t / (g.tiles_x * g.tiles_y);
    float y[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = 2 * a2 + i, ox = 2 * b2 + j;
            y[i][j] = (oy < g.Ho && ox < g.Wo) ? dy[((size_t)(img * g.Ho + oy) * g.Wo + ox) * ldy + n] : 0.f;
        }
    float s[4][2];                          // A along rows: (y0, y0 + y1, y0 - y1, -y1)
#pragma unroll
    for (int j = 0; j < 2; ++j) { s[0][j] = y[0][j]; s[1][j] = y[0][j] + y[1][j]; s[2][j] = y[0][j] - y[1][j]; s[3][j] = -y[1][j]; }
    float* dst = Dv + (size_t)t * g.N + n;
    const size_t bs = (size_t)g.M * g.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[0] = s[i][0]; dst += bs; GDN_KEEP(dst);
        dst[0] = s[i][0] + s[i][1]; dst += bs; GDN_KEEP(dst);
        dst[0] = s[i][0] - s[i][1]; dst += bs; GDN_KEEP(dst);
        dst[0] = -s[i][1]; dst += bs; GDN_KEEP(dst);
    }
}

// U = G g G^T.  swap = 0: U[bin][n][c] from w[tap][n][c] (forward);  swap = 1: U[bin][c][n] from the flipped taps
// (data gradient: correlation of dy with w[n][c][2 - ty][2 - tx], output channel c)
// Uswap != NULL (forward of a layer that will run backward): the data gradient's set (taps flipped: bins permuted
// (3,1,2,0) per axis; channel roles swapped) is written by the same launch from the same nine reads -- the weights do not
// change between a step's forward and backward, so the backward needs no weight transform of its own.
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int N, int C, int swap,
                                                           float* __restrict__ Uswap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int c = i % C, n = i / C;
    float gk[3][3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const int tap = swap ? (2 - ty) * 3 + (2 - tx) : ty * 3 + tx;
            gk[ty][tx] = w[((size_t)tap * N + n) * C + c];
        }
    float r[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        r[0][j] = gk[0][j];
        r[1][j] = 0.5f * (gk[0][j] + gk[1][j] + gk[2][j]);
        r[2][j] = 0.5f * (gk[0][j] - gk[1][j] + gk[2][j]);
        r[3][j] = gk[2][j];
    }
    float* dst = U ? U + (swap ? (size_t)c * N + n : (size_t)n * C + c) : nullptr;      // (U == NULL: only the swapped set)
    const size_t bs = (size_t)N * C;
    float u[4][4];
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2) {
        u[i2][0] = r[i2][0];
        u[i2][1] = 0.5f * (r[i2][0] + r[i2][1] + r[i2][2]);
        u[i2][2] = 0.5f * (r[i2][0] - r[i2][1] + r[i2][2]);
        u[i2][3] = r[i2][2];
    }
    if (dst) {
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) { dst[0] = u[i2][j2]; dst += bs; GDN_KEEP(dst); }
    }
    if (Uswap) {
        // G flip(g) G^T = P (G g G^T) P with P the permutation (3,1,2,0): flipping the taps swaps rows 0 <-> 3 of G g
        float* d2 = Uswap + (size_t)c * N + n;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const int pi = i2 == 0 ? 3 : (i2 == 3 ? 0 : i2), pj = j2 == 0 ? 3 : (j2 == 3 ? 0 : j2);
                d2[0] = u[pi][pj]; d2 += bs; GDN_KEEP(d2);
            }
    }
}

// The same transform written as bf16 x 3 packed panels (gemm_x3.h) for the GEMMs that run on the bf16 matrix pipe.
// swap = 0: rows = n (output channel), k = c -- the forward's B operand U[bin][n][c];  swap = 1: rows = c, k = n from the
// flipped taps -- the data gradient's.  thread = (row, 8 consecutive k): 9 taps x 8 values in, 16 bins x 3 planes x 16 bytes out.
// grid.y selects the set: y = 0 -> (Up, swap), y = 1 -> (Up1, swap1): a forward that will run backward writes both in one launch.
__global__ __launch_bounds__(256) void wino_weights_x3_kernel(const float* __restrict__ w, unsigned char* __restrict__ Up0, int N, int C,
                                                              int swap0, unsigned char* __restrict__ Up1, int swap1) {
    unsigned char* __restrict__ Up = blockIdx.y ? Up1 : Up0;
    const int swap = blockIdx.y ? swap1 : swap0;
    const int rows = swap ? C : N, K = swap ? N : C, k8n = K / 8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * k8n) return;
    // a wave = 16 consecutive rows x the 4 chunks of ONE k block: 1 KB contiguous per (bin, plane) store in the packed layout;
    // its loads are 16 runs of 128 B (k = c contiguous) or 8 x 4 runs of 64 B (swap: k = n, stride C)
    const int row = (i >> 2) % rows, k8 = (i & 3) + 4 * ((i >> 2) / rows);
    float gk[8][3][3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const int tap = swap ? (2 - ty) * 3 + (2 - tx) : ty * 3 + tx;
            if (swap) {
#pragma unroll
                for (int e = 0; e < 8; ++e) gk[e][ty][tx] = w[((size_t)tap * N + k8 * 8 + e) * C + row];
            } else {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(w + ((size_t)tap * N + row) * C + k8 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(w + ((size_t)tap * N + row) * C + k8 * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { gk[e][ty][tx] = lo[e]; gk[4 + e][ty][tx] = hi[e]; }
            }
        }
    const int KB = K / X3_BK;
    const size_t per_bin = x3_packed_bytes(rows, K);
    unsigned char* dst = Up + x3_off(row, k8 * 8, 0, KB);
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
            unsigned h[3][4];
            float uu[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // (G g G^T)[i2][j2]: rows of G = (1,0,0), (.5,.5,.5), (.5,-.5,.5), (0,0,1)
                float r[3];
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    r[j] = i2 == 0 ? gk[e][0][j] : i2 == 3 ? gk[e][2][j]
                         : 0.5f * (i2 == 1 ? gk[e][0][j] + gk[e][1][j] + gk[e][2][j] : gk[e][0][j] - gk[e][1][j] + gk[e][2][j]);
                uu[e] = j2 == 0 ? r[0] : j2 == 3 ? r[2] : 0.5f * (j2 == 1 ? r[0] + r[1] + r[2] : r[0] - r[1] + r[2]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) x3_split2(uu[2 * q], uu[2 * q + 1], h[0][q], h[1][q], h[2][q]);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint4 v = {h[p][0], h[p][1], h[p][2], h[p][3]};
                *reinterpret_cast<uint4*>(dst + (size_t)(i2 * 4 + j2) * per_bin + (size_t)p * 128 * 64) = v;
            }
        }
}

// y = A^T m A (2x2 outputs per tile) with the conv_igemm epilogue; stats slot = group of 4 tiles
// bnb_y != NULL (data gradient): y is the gradient of a train-mode BatchNorm's output z = [relu](BN(bnb_y)); `stats` then
// receives that layer's backward partials (sum dz, sum dz*xhat per slot; bnb_co = [scale, shift, mean, invstd][Nout])
// instead of sum / sum of squares, so the stand-alone reduce pass over (dy, y) disappears.
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mo, float* __restrict__ y, int ldy,
                                                          const float* __restrict__ addsrc, int ld_add,
                                                          float* __restrict__ stats, const float* __restrict__ ep_scale,
                                                          const float* __restrict__ ep_shift, int act, WinoGeom g, int Nout,
                                                          int q_shift, const float* __restrict__ bnb_y, int ld_bnb,
                                                          const float* __restrict__ bnb_co, int bnb_relu) {
    __shared__ float red[256 * 2];
    int t, n;
    const bool live = wino_decode(g.M, q_shift, t, n);
    float s1 = 0.f, s2 = 0.f;
    if (live) {
        const int b2 = t % g.tiles_x, a2 = (t / g.tiles_x) % g.tiles_y, img = t / (g.tiles_x * g.tiles_y);
        float m[4][4];
        const float* src = Mo + (size_t)t * Nout + n;
        const size_t bs = (size_t)g.M * Nout;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { m[i][j] = *src; src += bs; GDN_KEEP(src); }
        float r[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[0][j] = m[0][j] + m[1][j] + m[2][j]; r[1][j] = m[1][j] - m[2][j] - m[3][j]; }
        const float es = ep_scale ? ep_scale[n] : 1.f, et = ep_shift ? ep_shift[n] : 0.f;
        float bsc = 0.f, bt = 0.f, bmu = 0.f, bis = 0.f;
        if (bnb_y) { bsc = bnb_co[n]; bt = bnb_co[Nout + n]; bmu = bnb_co[2 * Nout + n]; bis = bnb_co[3 * Nout + n]; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float o[2] = {r[i][0] + r[i][1] + r[i][2], r[i][1] - r[i][2] - r[i][3]};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = 2 * a2 + i, ox = 2 * b2 + j;
                if (oy < g.Ho && ox < g.Wo) {
                    float val = o[j];
                    if (!bnb_y) { s1 += val; s2 += val * val; }
                    if (ep_scale) val = val * es + et;
                    if (act & GDN_ACT_RELU) val = fmaxf(val, 0.f);
                    const size_t px = (size_t)(img * g.Ho + oy) * g.Wo + ox;
                    if (addsrc) val += addsrc[px * ld_add + n];
                    if (act & GDN_ACT_TANH) val = tanhf(val);
                    y[px * ldy + n] = val;
                    if (bnb_y) {
                        const float yv = bnb_y[px * ld_bnb + n];
                        float dz = val;
                        if (bnb_relu && !(yv * bsc + bt > 0.f)) dz = 0.f;
                        s1 += dz; s2 += dz * ((yv - bmu) * bis);
                    }
                }
            }
        }
    }
    if (stats) {
        red[threadIdx.x * 2] = s1; red[threadIdx.x * 2 + 1] = s2;
        __syncthreads();
        if (threadIdx.x < 64) {
            float a1 = 0.f, a2s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a1 += red[(j * 64 + threadIdx.x) * 2]; a2s += red[(j * 64 + threadIdx.x) * 2 + 1]; }
            const int slot = blockIdx.x >> q_shift;
            const int ch = (blockIdx.x & ((1 << q_shift) - 1)) * 64 + threadIdx.x;
            stats[((size_t)slot * 2 + 0) * Nout + ch] = a1;
            stats[((size_t)slot * 2 + 1) * Nout + ch] = a2s;
        }
    }
}

// dW[tap][n][c] = (G^T P G)[ty][tx]
__global__ __launch_bounds__(256) void wino_wgrad_output_kernel(const float* __restrict__ P, float* __restrict__ dw, int N, int C,
                                                                int nsplit) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    float p[4][4];
    const float* src = P + i;
    const size_t bs = (size_t)N * C;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float v = *src;
            for (int sp = 1; sp < nsplit; ++sp) v += src[(size_t)sp * WINO_BINS * bs];      // split reduction, fixed order
            p[a][b] = v; src += bs; GDN_KEEP(src);
        }
    float r[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        r[0][b] = p[0][b] + 0.5f * (p[1][b] + p[2][b]);
        r[1][b] = 0.5f * (p[1][b] - p[2][b]);
        r[2][b] = 0.5f * (p[1][b] + p[2][b]) + p[3][b];
    }
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
        dw[(size_t)(ty * 3 + 0) * bs + i] = r[ty][0] + 0.5f * (r[ty][1] + r[ty][2]);
        dw[(size_t)(ty * 3 + 1) * bs + i] = 0.5f * (r[ty][1] - r[ty][2]);
        dw[(size_t)(ty * 3 + 2) * bs + i] = 0.5f * (r[ty][1] + r[ty][2]) + r[ty][3];
    }
}


// ---- F(4x4, 3x3): 6x6 patches, 36 bins, 2.25 multiplications per output instead of 4 (and 2.25 transformed values per pixel
// instead of 4).  Interpolation points 0, +-a, +-b, inf with a = 5/8, b = 3/2 -- NOT the textbook 0, +-1, +-2: in fp32 (transforms
// rounded, products exact: what the bf16 x 3 GEMMs deliver) the textbook set has 2.0x the rms and 4x the worst-case error of this
// one, whose rms is 1.25x the direct fp32 sum's (numpy model, tests/test_winoconv_model_cpu.py; a scan of dyadic pairs is flat
// around a ~ 0.6-0.7, b ~ 1.5, i.e. a b ~ 1).  Both values are dyadic, so every power and every entry of B^T is exact in fp32.
// With M(x) = x (x^2 - a^2)(x^2 - b^2):  B^T rows = coefficients of M(x) / (x - p) (row inf: M itself),  G rows = (1, p, p^2) / N_p
// with N_p = prod_{q != p} (p - q) (row inf: (0, 0, 1)),  A^T columns = (1, p, p^2, p^3) (column inf: (0, 0, 0, 1)).
// The weight gradient is F(3x3,4x4) on the same points: the saved B^T d B is reused, dy takes G_w (6x4: rows (1, p, p^2, p^3) / N_p),
// the reduction over tiles is the TN GEMM, and A_w^T (3x6: columns (1, p, p^2), column inf (0, 0, 1)) folds the 36 bins into 9 taps.
// F(2x2,3x3) stays for reflection-padded layers, small images and the fp32-matrix-pipe fallback.
#define W4_A 0.625f
#define W4_B 1.5f
constexpr float W4_A2 = W4_A * W4_A, W4_B2 = W4_B * W4_B, W4_A3 = W4_A2 * W4_A, W4_B3 = W4_B2 * W4_B;
constexpr float W4_P = W4_A2 * W4_B2, W4_S = W4_A2 + W4_B2;                                       // a^2 b^2, a^2 + b^2
constexpr float W4_I0 = (float)(1.0 / ((double)W4_A2 * W4_B2));                                  // 1 / N_0
constexpr float W4_IA = (float)(1.0 / (2.0 * W4_A2 * ((double)W4_A2 - W4_B2)));                  // 1 / N_a = 1 / N_-a
constexpr float W4_IB = (float)(1.0 / (2.0 * W4_B2 * ((double)W4_B2 - W4_A2)));                  // 1 / N_b = 1 / N_-b
__device__ __forceinline__ void w4_bt(const float (&d)[6], float (&o)[6]) {
    const float ea = d[4] - W4_B2 * d[2], oa = W4_A * (d[3] - W4_B2 * d[1]);
    const float eb = d[4] - W4_A2 * d[2], ob = W4_B * (d[3] - W4_A2 * d[1]);
    o[0] = W4_P * d[0] - W4_S * d[2] + d[4];
    o[1] = ea + oa;
    o[2] = ea - oa;
    o[3] = eb + ob;
    o[4] = eb - ob;
    o[5] = W4_P * d[1] - W4_S * d[3] + d[5];
}
__device__ __forceinline__ void w4_g(const float (&g)[3], float (&o)[6]) {        // G (6x3)
    const float ea = W4_IA * (g[0] + W4_A2 * g[2]), oa = (W4_IA * W4_A) * g[1];
    const float eb = W4_IB * (g[0] + W4_B2 * g[2]), ob = (W4_IB * W4_B) * g[1];
    o[0] = W4_I0 * g[0];
    o[1] = ea + oa;
    o[2] = ea - oa;
    o[3] = eb + ob;
    o[4] = eb - ob;
    o[5] = g[2];
}
__device__ __forceinline__ void w4_gw(const float (&y)[4], float (&o)[6]) {       // G_w (6x4)
    const float ea = W4_IA * (y[0] + W4_A2 * y[2]), oa = (W4_IA * W4_A) * (y[1] + W4_A2 * y[3]);
    const float eb = W4_IB * (y[0] + W4_B2 * y[2]), ob = (W4_IB * W4_B) * (y[1] + W4_B2 * y[3]);
    o[0] = W4_I0 * y[0];
    o[1] = ea + oa;
    o[2] = ea - oa;
    o[3] = eb + ob;
    o[4] = eb - ob;
    o[5] = y[3];
}
__device__ __forceinline__ void w4_at(const float (&m)[6], float (&o)[4]) {       // A^T (4x6)
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = m[0] + s12 + s34;
    o[1] = W4_A * d12 + W4_B * d34;
    o[2] = W4_A2 * s12 + W4_B2 * s34;
    o[3] = W4_A3 * d12 + W4_B3 * d34 + m[5];
}
__device__ __forceinline__ void w4_awt(const float (&p)[6], float (&o)[3]) {      // A_w^T (3x6)
    const float s12 = p[1] + p[2], d12 = p[1] - p[2], s34 = p[3] + p[4], d34 = p[3] - p[4];
    o[0] = p[0] + s12 + s34;
    o[1] = W4_A * d12 + W4_B * d34;
    o[2] = W4_A2 * s12 + W4_B2 * s34 + p[5];
}

// V = B^T d B of the 6x6 patch whose top-left corner is (4a - pad_off, 4b - pad_off); in_scale / up2x as wino_input_kernel
// (zero padding only: reflection-padded layers stay on F(2x2,3x3))
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, int ldx, float* __restrict__ V, WinoGeom g,
                                                          const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                          int in_relu, int up2x) {
    int t, c;
    if (!wino_decode(g.M, g.cq_shift, t, c)) return;
    const float is = in_scale ? in_scale[c] : 1.f, it = in_scale ? in_shift[c] : 0.f;
    const float lo = in_relu ? 0.f : -3.402823466e38f;
    const int b2 = t % g.tiles_x, a2 = (t / g.tiles_x) % g.tiles_y, img = t / (g.tiles_x * g.tiles_y);
    float r[6][6];                           // B^T along rows, one patch column at a time
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int ix = 4 * b2 - g.pad_off + j;
        const bool col_ok = ix >= 0 && ix < g.W;
        float d[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int iy = 4 * a2 - g.pad_off + i;
            const bool ok = col_ok && iy >= 0 && iy < g.H;
            float v = 0.f;
            if (ok) {
                if (up2x) {
                    // x is the LOW-resolution tensor [B][H/2][W/2][ldx]: the x2 bilinear upsampling happens here (up2x.h)
                    const int Hl = g.H >> 1, Wl = g.W >> 1;
                    float ly, lx; int y0, y1, x0, x1;
                    up_src(iy, Hl, up2x - 1, ly, y0, y1);
                    up_src(ix, Wl, up2x - 1, lx, x0, x1);
                    v = up2x_at(x + ((size_t)(img * Hl + y0) * Wl) * ldx + c, x + ((size_t)(img * Hl + y1) * Wl) * ldx + c, ldx, ly, x0, x1, lx);
                } else {
                    v = x[((size_t)(img * g.H + iy) * g.W + ix) * ldx + c];
                    if (in_scale) v = fmaxf(v * is + it, lo);
                }
            }
            d[i] = v;
        }
        float o[6];
        w4_bt(d, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = o[i];
    }
    float* dst = V + (size_t)t * g.C + c;
    const size_t bs = (size_t)g.M * g.C;
#pragma unroll
    for (int i = 0; i < 6; ++i) {            // ... and along columns
        float o[6];
        w4_bt(r[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) { dst[0] = o[j]; dst += bs; GDN_KEEP(dst); }
    }
}

// Dv = G_w dy G_w^T of the 4x4 output tile (weight gradient)
__global__ __launch_bounds__(256) void wino4_dy_kernel(const float* __restrict__ dy, int ldy, float* __restrict__ Dv, WinoGeom g) {
    int t, n;
    if (!wino_decode(g.M, g.nq_shift, t, n)) return;
    const int b2 = t % g.tiles_x, a2 = (t / g.tiles_x) % g.tiles_y, img = t / (g.tiles_x * g.tiles_y);
    float s[6][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ox = 4 * b2 + j;
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oy = 4 * a2 + i;
            y[i] = (oy < g.Ho && ox < g.Wo) ? dy[((size_t)(img * g.Ho + oy) * g.Wo + ox) * ldy + n] : 0.f;
        }
        float o[6];
        w4_gw(y, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i][j] = o[i];
    }
    float* dst = Dv + (size_t)t * g.N + n;
    const size_t bs = (size_t)g.M * g.N;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float o[6];
        w4_gw(s[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) { dst[0] = o[j]; dst += bs; GDN_KEEP(dst); }
    }
}

// U = G g G^T as bf16 x 3 packed panels, 36 bins (arguments as wino_weights_x3_kernel)
__global__ __launch_bounds__(256) void wino4_weights_x3_kernel(const float* __restrict__ w, unsigned char* __restrict__ Up0, int N, int C,
                                                               int swap0, unsigned char* __restrict__ Up1, int swap1) {
    unsigned char* __restrict__ Up = blockIdx.y ? Up1 : Up0;
    const int swap = blockIdx.y ? swap1 : swap0;
    const int rows = swap ? C : N, K = swap ? N : C, k8n = K / 8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * k8n) return;
    // a wave = 16 consecutive rows x the 4 chunks of ONE k block: 1 KB contiguous per (bin, plane) store in the packed layout
    // (thread = (row, consecutive k8) wrote sixteen 64-byte pieces 24 KB apart per store: 54 us for 113 MB at 512 channels)
    const int row = (i >> 2) % rows, k8 = (i & 3) + 4 * ((i >> 2) / rows);
    float r[8][6][3];                        // G along the tap rows, per k value
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
        float gk[8][3];
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const int tap = swap ? (2 - ty) * 3 + (2 - tx) : ty * 3 + tx;
            if (swap) {
#pragma unroll
                for (int e = 0; e < 8; ++e) gk[e][ty] = w[((size_t)tap * N + k8 * 8 + e) * C + row];
            } else {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(w + ((size_t)tap * N + row) * C + k8 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(w + ((size_t)tap * N + row) * C + k8 * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { gk[e][ty] = lo[e]; gk[4 + e][ty] = hi[e]; }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o[6];
            w4_g(gk[e], o);
#pragma unroll
            for (int i2 = 0; i2 < 6; ++i2) r[e][i2][tx] = o[i2];
        }
    }
    const int KB = K / X3_BK;
    const size_t per_bin = x3_packed_bytes(rows, K);
    unsigned char* dst = Up + x3_off(row, k8 * 8, 0, KB);
#pragma unroll
    for (int i2 = 0; i2 < 6; ++i2) {
        float uu[8][6];
#pragma unroll
        for (int e = 0; e < 8; ++e) w4_g(r[e][i2], uu[e]);
#pragma unroll
        for (int j2 = 0; j2 < 6; ++j2) {
            unsigned h[3][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x3_split2(uu[2 * q][j2], uu[2 * q + 1][j2], h[0][q], h[1][q], h[2][q]);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint4 v = {h[p][0], h[p][1], h[p][2], h[p][3]};
                *reinterpret_cast<uint4*>(dst + (size_t)(i2 * 6 + j2) * per_bin + (size_t)p * 128 * 64) = v;
            }
        }
    }
}

// y = A^T m A (4x4 outputs per tile); epilogue, BatchNorm partials and the data gradient's BatchNorm-backward partials exactly as
// wino_output_kernel (stats slot = group of 4 tiles)
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ Mo, float* __restrict__ y, int ldy,
                                                           const float* __restrict__ addsrc, int ld_add,
                                                           float* __restrict__ stats, const float* __restrict__ ep_scale,
                                                           const float* __restrict__ ep_shift, int act, WinoGeom g, int Nout,
                                                           int q_shift, const float* __restrict__ bnb_y, int ld_bnb,
                                                           const float* __restrict__ bnb_co, int bnb_relu) {
    __shared__ float red[256 * 2];
    int t, n;
    const bool live = wino_decode(g.M, q_shift, t, n);
    float s1 = 0.f, s2 = 0.f;
    if (live) {
        const int b2 = t % g.tiles_x, a2 = (t / g.tiles_x) % g.tiles_y, img = t / (g.tiles_x * g.tiles_y);
        float r[4][6];                       // A^T along rows, one bin column at a time
        const float* src = Mo + (size_t)t * Nout + n;
        const size_t bs = (size_t)g.M * Nout;
        {
            float m[6][6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) { m[i][j] = *src; src += bs; GDN_KEEP(src); }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float col[6] = {m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]};
                float o[4];
                w4_at(col, o);
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i][j] = o[i];
            }
        }
        const float es = ep_scale ? ep_scale[n] : 1.f, et = ep_shift ? ep_shift[n] : 0.f;
        float bsc = 0.f, bt = 0.f, bmu = 0.f, bis = 0.f;
        if (bnb_y) { bsc = bnb_co[n]; bt = bnb_co[Nout + n]; bmu = bnb_co[2 * Nout + n]; bis = bnb_co[3 * Nout + n]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o[4];
            w4_at(r[i], o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int oy = 4 * a2 + i, ox = 4 * b2 + j;
                if (oy < g.Ho && ox < g.Wo) {
                    float val = o[j];
                    if (!bnb_y) { s1 += val; s2 += val * val; }
                    if (ep_scale) val = val * es + et;
                    if (act & GDN_ACT_RELU) val = fmaxf(val, 0.f);
                    const size_t px = (size_t)(img * g.Ho + oy) * g.Wo + ox;
                    if (addsrc) val += addsrc[px * ld_add + n];
                    if (act & GDN_ACT_TANH) val = tanhf(val);
                    y[px * ldy + n] = val;
                    if (bnb_y) {
                        const float yv = bnb_y[px * ld_bnb + n];
                        float dz = val;
                        if (bnb_relu && !(yv * bsc + bt > 0.f)) dz = 0.f;
                        s1 += dz; s2 += dz * ((yv - bmu) * bis);
                    }
                }
            }
        }
    }
    if (stats) {
        red[threadIdx.x * 2] = s1; red[threadIdx.x * 2 + 1] = s2;
        __syncthreads();
        if (threadIdx.x < 64) {
            float a1 = 0.f, a2s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a1 += red[(j * 64 + threadIdx.x) * 2]; a2s += red[(j * 64 + threadIdx.x) * 2 + 1]; }
            const int slot = blockIdx.x >> q_shift;
            const int ch = (blockIdx.x & ((1 << q_shift) - 1)) * 64 + threadIdx.x;
            stats[((size_t)slot * 2 + 0) * Nout + ch] = a1;
            stats[((size_t)slot * 2 + 1) * Nout + ch] = a2s;
        }
    }
}

// dW[tap][n][c] = (A_w^T P A_w)[ty][tx], the split partial products summed in order
__global__ __launch_bounds__(256) void wino4_wgrad_output_kernel(const float* __restrict__ P, float* __restrict__ dw, int N, int C,
                                                                 int nsplit) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const float* src = P + i;
    const size_t bs = (size_t)N * C;
    float r[3][6];
    {
        float p[6][6];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                float v = *src;
                for (int sp = 1; sp < nsplit; ++sp) v += src[(size_t)sp * 36 * bs];
                p[a][b] = v; src += bs; GDN_KEEP(src);
            }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const float col[6] = {p[0][b], p[1][b], p[2][b], p[3][b], p[4][b], p[5][b]};
            float o[3];
            w4_awt(col, o);
#pragma unroll
            for (int a = 0; a < 3; ++a) r[a][b] = o[a];
        }
    }
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
        float o[3];
        w4_awt(r[ty], o);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) dw[(size_t)(ty * 3 + tx) * bs + i] = o[tx];
    }
}

bool wino_geom(const gdn_conv_geom* g, WinoGeom& f) {
    if (!g || g->transposed || g->stride != 1 || g->k != 3 || g->pad != 1) return false;
    if (g->pad_mode == 1 && (g->H < 4 || g->W < 4)) return false;        // mirrored rows 1 and H-2 must be distinct interior rows
    if ((g->Cin % 64) || (g->Cout % 64) || g->Cin > 512 || g->Cout > 512) return false;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (!pow2(g->Cin / 64) || !pow2(g->Cout / 64)) return false;
    f.B = g->B; f.H = g->H; f.W = g->W; f.C = g->Cin; f.N = g->Cout;
    f.Ho = g->H; f.Wo = g->W; f.pad_off = 1; f.reflect = g->pad_mode == 1;
    f.tiles_y = cdiv(g->H, 2); f.tiles_x = cdiv(g->W, 2);
    f.M = g->B * f.tiles_y * f.tiles_x;
    f.cq_shift = 0; while ((64 << f.cq_shift) < f.C) ++f.cq_shift;
    f.nq_shift = 0; while ((64 << f.nq_shift) < f.N) ++f.nq_shift;
    f.x3 = (g->hints & GDN_HINT_NO_X3) ? 0 : 1;
    f.T = 2; f.bins = WINO_BINS;
    // F(4x4,3x3) where every GEMM of the layer runs as bf16 x 3 split products (the transformed weights exist as panels only),
    // the border is zero padding, and rounding the image up to whole 4 x 4 tiles computes at most 15 % more pixels.
    // GDN_HINT_NO_WINO_F4 in the geometry keeps F(2x2,3x3) (A/B measurements, accuracy studies; gdn_amd/ops.py: set_wino_f4).  Like
    // the bf16 x 3 switch it is part of the geometry -- no environment read -- so the forward that lays out the saved state and the
    // backward that reads it agree on the plan by construction.
    if (f.x3 && !f.reflect && g->H >= 4 && g->W >= 4 && !(g->hints & GDN_HINT_NO_WINO_F4)) {
        const int ty = cdiv(g->H, 4), tx = cdiv(g->W, 4), M4 = g->B * ty * tx;
        const bool fits = (int64_t)ty * tx * 16 * 100 <= (int64_t)g->H * g->W * 115;
        if (fits && gemm_x3_ok(M4, f.N, f.C) && gemm_x3_ok(M4, f.C, f.N) && gemm_x3_tn_ok(M4, f.N, f.C)) {
            f.T = 4; f.bins = 36; f.tiles_y = ty; f.tiles_x = tx; f.M = M4;
        }
    }
    return true;
}

inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }
inline size_t v_bytes(const WinoGeom& f) { return al256((size_t)f.bins * f.M * f.C * 4); }
// transformed weights: fp32 [16][N][C], or the bf16 x 3 panels of either orientation (rows padded to whole 128-row tiles)
inline size_t u_bytes(const WinoGeom& f) {
    size_t b = (size_t)f.bins * f.N * f.C * 4;
    const size_t p1 = (size_t)f.bins * x3_packed_bytes(f.N, f.C), p2 = (size_t)f.bins * x3_packed_bytes(f.C, f.N);
    if (p1 > b) b = p1;
    if (p2 > b) b = p2;
    return al256(b);
}
// GDN_HINT_NO_X3 in the geometry keeps every per-bin GEMM on the fp32 MFMA (ranks that share a GPU, A/B measurements, accuracy
// studies).  It is part of the layer's geometry, so the forward that writes the saved weight set and the backward that reads
// it agree on its form by construction.
inline size_t m_bytes(const WinoGeom& f) { return al256((size_t)f.bins * f.M * f.N * 4); }
inline bool tn_x3(const WinoGeom& f) { return f.x3 && gemm_x3_tn_ok(f.M, f.N, f.C); }
inline int tn_splits(const WinoGeom& f) { return tn_x3(f) ? gemm_x3_tn_splits(f.bins, f.M, f.N, f.C) : wino_tn_splits(f.M, f.N, f.C); }
// (workspace sizing: the larger of the two kernels' split counts)
inline int tn_splits_max(const WinoGeom& f) {
    const int a = wino_tn_splits(f.M, f.N, f.C), b = gemm_x3_tn_ok(f.M, f.N, f.C) ? gemm_x3_tn_splits(f.bins, f.M, f.N, f.C) : 1;
    return a > b ? a : b;
}

}  // namespace

// saved state of one forward for its backward: the transformed input V, then the data gradient's transformed weights
extern "C" size_t gdn_winoconv_state_bytes(const gdn_conv_geom* g) {
    WinoGeom f;
    return wino_geom(g, f) ? v_bytes(f) + u_bytes(f) : 0;
}

// workspace: V, U, Mo
extern "C" size_t gdn_winoconv_fwd_workspace_bytes(const gdn_conv_geom* g) {
    WinoGeom f;
    return wino_geom(g, f) ? v_bytes(f) + u_bytes(f) + m_bytes(f) : 0;
}

extern "C" int64_t gdn_winoconv_stats_slots(const gdn_conv_geom* g) {
    WinoGeom f;
    if (!wino_geom(g, f)) return GDN_ERR_UNSUPPORTED;
    return cdiv(f.M, 4);
}

extern "C" int gdn_winoconv_fwd(const gdn_conv_geom* g, const float* x, int32_t ldx, const float* w, float* y, int32_t ldy,
                                const float* addsrc, int32_t ld_add, float* stats, const float* ep_scale,
                                const float* ep_shift, int32_t act, const float* in_scale, const float* in_shift,
                                int32_t in_relu, int32_t in_up2x, void* state_out, void* workspace,
                                size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    WinoGeom f;
    if (!wino_geom(g, f)) return GDN_ERR_UNSUPPORTED;
    if (!x || !w || !y || (!ep_scale) != (!ep_shift) || (!in_scale) != (!in_shift)) return GDN_ERR_BAD_ARG;
    if (in_up2x < 0 || in_up2x > 2 || (in_up2x && (in_scale || (g->H & 1) || (g->W & 1)))) return GDN_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gdn_winoconv_fwd_workspace_bytes(g)) return GDN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)workspace;
    float* V = (float*)p; p += v_bytes(f);
    float* U = (float*)p; p += u_bytes(f);
    float* Mo = (float*)p;
    if (state_out) V = (float*)state_out;
    float* Usw = state_out ? (float*)((char*)state_out + v_bytes(f)) : nullptr;
    if (f.T == 4) {
        // F(4x4,3x3): every GEMM of the layer is a bf16 x 3 one (wino_geom); the data gradient's weight set rides along as above
        hipLaunchKernelGGL(wino4_input_kernel, dim3(cdiv(f.M, 4) << f.cq_shift), dim3(256), 0, st, x, ldx, V, f, in_scale, in_shift,
                           in_relu, in_up2x);
        hipLaunchKernelGGL(wino4_weights_x3_kernel, dim3(cdiv(f.N * f.C / 8, 256), Usw ? 2 : 1), dim3(256), 0, st, w, (unsigned char*)U, f.N,
                           f.C, 0, (unsigned char*)Usw, 1);
        launch_gemm_x3_nt((const float*)V, U, Mo, f.bins, f.M, f.N, f.C, st);
        hipLaunchKernelGGL(wino4_output_kernel, dim3(cdiv(f.M, 4) << f.nq_shift), dim3(256), 0, st, (const float*)Mo, y, ldy, addsrc,
                           ld_add, stats, ep_scale, ep_shift, act, f, f.N, f.nq_shift, (const float*)nullptr, 0,
                           (const float*)nullptr, 0);
        return gdn_launch_status();
    }
    hipLaunchKernelGGL(wino_input_kernel, dim3(cdiv(f.M, 4) << f.cq_shift), dim3(256), 0, st, x, ldx, V, f, in_scale, in_shift,
                       in_relu, in_up2x);
    // the per-bin GEMMs run as bf16 x 3 split products on the bf16 matrix pipe where the shape allows (gemm_x3.h); the data
    // gradient's weight set (its GEMM has N = Cin) is written in the form ITS kernel will read
    const bool x3f = f.x3 && gemm_x3_ok(f.M, f.N, f.C), x3d = f.x3 && gemm_x3_ok(f.M, f.C, f.N);
    if (!x3f || (Usw && !x3d))
        hipLaunchKernelGGL(wino_weights_kernel, dim3(cdiv(f.N * f.C, 256)), dim3(256), 0, st, w, x3f ? (float*)nullptr : U, f.N,
                           f.C, 0, x3d ? (float*)nullptr : Usw);
    if (x3f && Usw && x3d)
        hipLaunchKernelGGL(wino_weights_x3_kernel, dim3(cdiv(f.N * f.C / 8, 256), 2), dim3(256), 0, st, w, (unsigned char*)U, f.N, f.C, 0,
                           (unsigned char*)Usw, 1);
    else if (x3f)
        hipLaunchKernelGGL(wino_weights_x3_kernel, dim3(cdiv(f.N * f.C / 8, 256)), dim3(256), 0, st, w, (unsigned char*)U, f.N, f.C, 0,
                           (unsigned char*)nullptr, 0);
    else if (Usw && x3d)
        hipLaunchKernelGGL(wino_weights_x3_kernel, dim3(cdiv(f.N * f.C / 8, 256)), dim3(256), 0, st, w, (unsigned char*)Usw, f.N, f.C, 1,
                           (unsigned char*)nullptr, 0);
    if (x3f) launch_gemm_x3_nt((const float*)V, U, Mo, WINO_BINS, f.M, f.N, f.C, st);
    else launch_wino_gemm((const float*)V, (const float*)U, Mo, f.M, f.N, f.C, st);
    hipLaunchKernelGGL(wino_output_kernel, dim3(cdiv(f.M, 4) << f.nq_shift), dim3(256), 0, st, (const float*)Mo, y, ldy, addsrc,
                       ld_add, stats, ep_scale, ep_shift, act, f, f.N, f.nq_shift, (const float*)nullptr, 0,
                       (const float*)nullptr, 0);
    return gdn_launch_status();
}

// workspace: Vd (transformed dy for the data gradient, [16][M][N]) / Dv (for the weight gradient, same size), U', Eo / P
extern "C" size_t gdn_winoconv_bwd_workspace_bytes(const gdn_conv_geom* g) {
    WinoGeom f;
    if (!wino_geom(g, f)) return 0;
    // reflection layers run the data gradient over the padded domain: more tiles, plus the padded gradient itself
    const size_t Md = f.reflect ? (size_t)f.B * cdiv(f.H + 2, 2) * cdiv(f.W + 2, 2) : (size_t)f.M;
    const size_t vd = al256((size_t)f.bins * Md * f.N * 4), eo = al256((size_t)f.bins * Md * f.C * 4);
    const size_t padded = f.reflect ? al256((size_t)f.B * (f.H + 2) * (f.W + 2) * f.C * 4) : 0;
    const size_t pr = (size_t)tn_splits_max(f) * al256((size_t)f.bins * f.N * f.C * 4);   // weight-gradient products, one set per split
    return (vd > m_bytes(f) ? vd : m_bytes(f)) + u_bytes(f) + (eo > pr ? eo : pr) + padded;
}

// slots of the BatchNorm-backward partials the data-gradient epilogue can emit (0: not available for this layer)
extern "C" int64_t gdn_winoconv_bnb_slots(const gdn_conv_geom* g) {
    WinoGeom f;
    if (!wino_geom(g, f) || f.reflect) return 0;
    return cdiv(f.M, 4);
}

extern "C" int gdn_winoconv_bwd(const gdn_conv_geom* g, const float* dy, int32_t ldy, const float* w, const void* state,
                                float* dx, int32_t ldx, const float* addsrc, int32_t ld_add, float* dw,
                                const float* bnb_y, int32_t ld_bnb, const float* bnb_co, int32_t bnb_relu,
                                float* bnb_partial, int32_t dx_up2x, void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    WinoGeom f;
    if (!wino_geom(g, f)) return GDN_ERR_UNSUPPORTED;
    if (!dy || (!dx && !dw) || (dx && !w && !state) || (dw && !state)) return GDN_ERR_BAD_ARG;
    if (bnb_y && (!dx || !bnb_co || !bnb_partial)) return GDN_ERR_BAD_ARG;
    if (bnb_y && f.reflect) return GDN_ERR_UNSUPPORTED;
    if (dx && f.reflect && ((ldx % 4) || (addsrc && (ld_add % 4)))) return GDN_ERR_UNSUPPORTED;
    if (dx_up2x < 0 || dx_up2x > 2 || (dx_up2x && ((g->H & 1) || (g->W & 1)))) return GDN_ERR_BAD_ARG;
    if (dx && dx_up2x && !f.reflect) return GDN_ERR_UNSUPPORTED;   // (the fold pass of a reflection layer carries the adjoint)
    if (!workspace || workspace_bytes < gdn_winoconv_bwd_workspace_bytes(g)) return GDN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)workspace;
    const size_t Md = f.reflect ? (size_t)f.B * cdiv(f.H + 2, 2) * cdiv(f.W + 2, 2) : (size_t)f.M;
    const size_t vd = al256((size_t)f.bins * Md * f.N * 4);
    float* Vd = (float*)p; p += (vd > m_bytes(f) ? vd : m_bytes(f));
    float* U = (float*)p; p += u_bytes(f);
    float* Eo = (float*)p;                       // data gradient: GEMM output [16][Md][C] (+ padded gradient); weight gradient: P
    if (f.T == 4) {
        if (dw) {
            hipLaunchKernelGGL(wino4_dy_kernel, dim3(cdiv(f.M, 4) << f.nq_shift), dim3(256), 0, st, dy, ldy, Vd, f);
            const int ns = tn_splits(f);
            launch_gemm_x3_tn((const float*)Vd, (const float*)state, Eo, f.bins, f.M, f.N, f.C, ns, st);
            hipLaunchKernelGGL(wino4_wgrad_output_kernel, dim3(cdiv(f.N * f.C, 256)), dim3(256), 0, st, (const float*)Eo, dw, f.N, f.C, ns);
        }
        if (dx) {
            WinoGeom fd = f;
            fd.C = f.N; fd.N = f.C; fd.cq_shift = f.nq_shift; fd.nq_shift = f.cq_shift;
            hipLaunchKernelGGL(wino4_input_kernel, dim3(cdiv(fd.M, 4) << fd.cq_shift), dim3(256), 0, st, dy, ldy, Vd, fd,
                               (const float*)nullptr, (const float*)nullptr, 0, 0);
            const float* Ud = U;
            if (state) Ud = (const float*)((const char*)state + v_bytes(f));      // transformed by the forward's launch
            else hipLaunchKernelGGL(wino4_weights_x3_kernel, dim3(cdiv(f.N * f.C / 8, 256)), dim3(256), 0, st, w, (unsigned char*)U, f.N, f.C, 1,
                                    (unsigned char*)nullptr, 0);
            launch_gemm_x3_nt((const float*)Vd, Ud, Eo, f.bins, fd.M, f.C, f.N, st);
            hipLaunchKernelGGL(wino4_output_kernel, dim3(cdiv(fd.M, 4) << fd.nq_shift), dim3(256), 0, st, (const float*)Eo, dx, ldx,
                               addsrc, ld_add, bnb_y ? bnb_partial : (float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0,
                               fd, f.C, fd.nq_shift, bnb_y, ld_bnb, bnb_co, bnb_relu);
        }
        return gdn_launch_status();
    }
    if (dw) {
        hipLaunchKernelGGL(wino_dy_kernel, dim3(cdiv(f.M, 4) << f.nq_shift), dim3(256), 0, st, dy, ldy, Vd, f);
        const int ns = tn_splits(f);
        if (tn_x3(f)) launch_gemm_x3_tn((const float*)Vd, (const float*)state, Eo, WINO_BINS, f.M, f.N, f.C, ns, st);
        else launch_wino_gemm_tn((const float*)Vd, (const float*)state, Eo, f.M, f.N, f.C, ns, st);
        hipLaunchKernelGGL(wino_wgrad_output_kernel, dim3(cdiv(f.N * f.C, 256)), dim3(256), 0, st, (const float*)Eo, dw, f.N, f.C, ns);
    }
    if (dx) {
        // the data gradient of a 3x3 layer is the same kind of convolution of dy with flipped, role-swapped taps: pad 1 onto
        // the H x W input for a zero-padded layer; pad 2 onto the (H+2) x (W+2) padded domain for a reflection-padded one,
        // whose border rows / columns are then folded back onto the rows they mirror
        WinoGeom fd = f;
        fd.C = f.N; fd.N = f.C; fd.cq_shift = f.nq_shift; fd.nq_shift = f.cq_shift; fd.reflect = 0;
        float* out = dx;
        int ld_out = ldx;
        if (f.reflect) {
            fd.Ho = f.H + 2; fd.Wo = f.W + 2; fd.pad_off = 2;
            fd.tiles_y = cdiv(fd.Ho, 2); fd.tiles_x = cdiv(fd.Wo, 2); fd.M = f.B * fd.tiles_y * fd.tiles_x;
            out = Eo + (size_t)WINO_BINS * fd.M * f.C;                 // padded-domain gradient, behind the GEMM output
            ld_out = f.C;
        }
        hipLaunchKernelGGL(wino_input_kernel, dim3(cdiv(fd.M, 4) << fd.cq_shift), dim3(256), 0, st, dy, ldy, Vd, fd,
                           (const float*)nullptr, (const float*)nullptr, 0, 0);
        const float* Ud = U;
        // (the forward decided the form of its saved set from the FORWARD tile count f.M; the padded-domain gradient of a
        // reflection layer has more tiles but the same N and K, so eligibility is the same)
        const bool x3d = f.x3 && gemm_x3_ok(f.M, f.C, f.N);
        if (state) Ud = (const float*)((const char*)state + v_bytes(f));      // transformed by the forward's launch
        else if (x3d) hipLaunchKernelGGL(wino_weights_x3_kernel, dim3(cdiv(f.N * f.C / 8, 256)), dim3(256), 0, st, w, (unsigned char*)U, f.N, f.C, 1,
                                         (unsigned char*)nullptr, 0);
        else hipLaunchKernelGGL(wino_weights_kernel, dim3(cdiv(f.N * f.C, 256)), dim3(256), 0, st, w, U, f.N, f.C, 1, (float*)nullptr);
        if (x3d) launch_gemm_x3_nt((const float*)Vd, Ud, Eo, WINO_BINS, fd.M, f.C, f.N, st);
        else launch_wino_gemm((const float*)Vd, Ud, Eo, fd.M, f.C, f.N, st);
        hipLaunchKernelGGL(wino_output_kernel, dim3(cdiv(fd.M, 4) << fd.nq_shift), dim3(256), 0, st, (const float*)Eo, out, ld_out,
                           f.reflect ? (const float*)nullptr : addsrc, ld_add, bnb_y ? bnb_partial : (float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, 0, fd, f.C, fd.nq_shift, bnb_y, ld_bnb, bnb_co, bnb_relu);
        if (f.reflect && dx_up2x) {
            // dx is the gradient of the LOW-resolution tensor the forward upsampled on load: fold + adjoint interpolation
            const int64_t nb = cdiv64((int64_t)f.B * (f.H / 2) * (f.W / 2) * (f.C / 4), 256);
            hipLaunchKernelGGL(reflect_fold_up2x_kernel, dim3((unsigned)(nb < 65536 * 8 ? nb : 65536 * 8)), dim3(256), 0, st,
                               (const void*)out, (void*)dx, ldx, (const void*)addsrc, ld_add, f.B, f.H, f.W, f.C, 1, dx_up2x - 1, 0);
        } else if (f.reflect) {
            const int64_t nb = cdiv64((int64_t)f.B * f.H * f.W * (f.C / 4), 256);
            hipLaunchKernelGGL(wino_reflect_fold_kernel, dim3((unsigned)(nb < 65536 * 8 ? nb : 65536 * 8)), dim3(256), 0, st,
                               (const float*)out, dx, ldx, addsrc, ld_add, f.B, f.H, f.W, f.C);
        }
    }
    return gdn_launch_status();
}

// Measurement hook (bench.py roofline, tools/): the 16 per-bin GEMMs of one forward alone, on already transformed operands
// V [16][M][Cin] and U [16][Cout][Cin]  ->  Mo [16][M][Cout].
extern "C" int gdn_winoconv_gemm(const gdn_conv_geom* g, const float* V, const float* U, float* Mo, void* stream) {
    (void)hipGetLastError();
    WinoGeom f;
    if (!wino_geom(g, f)) return GDN_ERR_UNSUPPORTED;
    if (!V || !U || !Mo) return GDN_ERR_BAD_ARG;
    launch_wino_gemm(V, U, Mo, f.B * cdiv(f.H, 2) * cdiv(f.W, 2), f.N, f.C, (hipStream_t)stream);      // (F(2x2,3x3) tiles whatever the layer's plan)
    return gdn_launch_status();
}
