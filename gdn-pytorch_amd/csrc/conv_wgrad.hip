// Weight-gradient kernel for gfx950 (MI355X), exact fp32 on the matrix cores.
//
//   dW[ky][kx][r][c] = sum_{b,oy,ox} G[b][oy][ox][r] * X[b][oy*s - p + ky][ox*s - p + kx][c]
//
// G is the "unshifted" tensor (the output gradient of a Conv2d, or the input of
// a ConvTranspose2d), X the "shifted" one.  The reduction runs over pixels, so
// it is a GEMM with M' = channels of G, N' = (kx, channels of X), K' = pixels:
//
//   * a workgroup owns one filter ROW ky, a 64-channel slab of G, a <=64-channel
//     slab of X, and a contiguous range of image rows (split-K over pixels);
//   * per 32-pixel row segment it stages G[32][64] and the X row segment
//     [(31*s + k)][slab] in LDS ONCE and reuses the X segment for all k taps of
//     the filter row -- the MFMA B operand of tap kx is the same LDS image read
//     at a shifted address (column jj = kx*slab + c lives at pix*s*slab + jj), so
//     X is fetched k times (once per filter row), not k*k times;
//   * accumulators: up to 9 tiles of 32x32 per wave (all kx of a 9x9 filter row);
//   * the per-split partial slabs are written with plain stores and summed by a
//     second kernel in a fixed order: bitwise reproducible, no atomics.
#include "common.h"

#define WG_ROWS 64        // channels of G per workgroup
#define WG_SLAB 64        // max channels of X per workgroup
#define WG_NTW_MAX 9      // max 32-column tiles per wave
// Pixels per row segment (the MFMA K' of one LDS image) are chosen per layer (even, <= TWMAX) so that
// segments tile the image width exactly: 32 for W=416, 52 for W in {208,104,52}, 26 for W=26.

struct WgradParams {
    const float* g; const float* x; float* part;
    int B, Hg, Wg, ldg, Cg;
    int Hx, Wx, ldx, Cx;
    int k, stride, pad, pad_mode;
    int cisl, n_cgt, n_cxt, S, rows_per_split;
    int tw;                         // pixels per row segment
    int g_bf16;                     // thin variant only: the G tensor holds bf16 (mixed-precision path)
    unsigned g_bytes, x_bytes;      // extents for the buffer descriptors (< 4 GiB each)
};

template <int WG_NTW, int TWMAX>
__global__ __launch_bounds__(256) void conv_wgrad_f32(const WgradParams p) {
    constexpr int MAXPOS = (TWMAX - 1) * 2 + 9;                 // stride <= 2, k <= 9
    constexpr int GV = TWMAX * WG_ROWS / 4 / 256;               // 16-B G loads per thread
    constexpr int XV = (MAXPOS * WG_SLAB / 4 + 255) / 256;      // 16-B X loads per thread
    __shared__ __attribute__((aligned(16))) float smem[TWMAX * WG_ROWS + MAXPOS * WG_SLAB + 64];
    float* Gs = smem;
    float* Xs = smem + TWMAX * WG_ROWS;
    // TWMAX == 32 is only launched with tw == 32: compile-time trip counts keep the 9-tile variant
    // at 2 waves/SIMD; every other segment width runs the TWMAX == 64 instantiation.
    const int WG_TW = (TWMAX == 32 && WG_NTW >= 7) ? 32 : p.tw;     // small-tile variants also take tw < 32 (half the LDS: one more wave/SIMD)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;
    int id = blockIdx.x;
    const int ky = id % p.k; id /= p.k;
    const int cgt = id % p.n_cgt; id /= p.n_cgt;
    const int cxt = id % p.n_cxt;
    const int split = id / p.n_cxt;

    const int CISL = p.cisl;
    const int ncols = p.k * CISL;
    const int NT = (ncols + 31) / 32;
    const int npos = (WG_TW - 1) * p.stride + p.k;
    const int cg0 = cgt * WG_ROWS, cx0 = cxt * CISL;
    const bool g_vec = (p.Cg % 4 == 0) && (p.ldg % 4 == 0);
    const bool x_vec = (CISL % 4 == 0) && (p.ldx % 4 == 0);

    f32x16 acc[WG_NTW];
#pragma unroll
    for (int t = 0; t < WG_NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int R = p.B * p.Hg;
    const int r0 = split * p.rows_per_split;
    const int r1 = min(R, r0 + p.rows_per_split);
    const int nseg = (p.Wg + WG_TW - 1) / WG_TW;
    const int total = (r1 - r0) * nseg;

    float rg[GV * 4 > 8 ? GV * 4 : 8], rx[XV * 4];

    // Per-thread constants of the vector load paths.  Tile loads are raw buffer loads: the lane part
    // is a constant 32-bit byte offset, the row/segment part rides in the scalar offset, and anything
    // that must read as zero (row tail, padding, channel tail) gets an out-of-range offset instead of
    // a branch -- VALU instructions compete with the fp32 MFMA for the same ALUs.
    const unsigned OOB = 0xFFFFFF00u;
    __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, (int)p.g_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    unsigned g_lane[GV]; int g_px[GV];
#pragma unroll
    for (int e = 0; e < GV; ++e) {
        const int idx = tid + 256 * e, c = (idx & 15) * 4;
        g_px[e] = idx >> 4;
        g_lane[e] = (cg0 + c < p.Cg) ? (unsigned)(g_px[e] * p.ldg + cg0 + c) * 4u : OOB;
    }
    const int q4 = CISL >> 2;
    unsigned x_lane[XV]; int x_pos[XV];
#pragma unroll
    for (int e = 0; e < XV; ++e) {
        const int idx = tid + 256 * e;
        const int pos = x_vec ? idx / q4 : 0, c = x_vec ? (idx - pos * q4) * 4 : 0;
        x_pos[e] = (x_vec && pos < npos) ? pos : -0x40000000;          // far out of any row
        x_lane[e] = (cx0 + c < p.Cx) ? (unsigned)(cx0 + c) * 4u : OOB;
    }

    auto gload = [&](int sidx) {
        const int r = r0 + sidx / nseg, ox0 = (sidx % nseg) * WG_TW;
        const int b = r / p.Hg, oy = r % p.Hg;
        int iy = oy * p.stride - p.pad + ky;
        if (p.pad_mode == 1) iy = reflect_idx(iy, p.Hx);
        const bool row_ok = iy >= 0 && iy < p.Hx;
        // ---- G tile: 32 pixels x 64 channels ----
        if (g_vec) {
            const int soff = (int)((unsigned)(((b * p.Hg + oy) * p.Wg + ox0) * p.ldg) * 4u);
            const int wrem = row_ok ? min(p.Wg - ox0, WG_TW) : 0;
#pragma unroll
            for (int e = 0; e < GV; ++e) {
                const unsigned vo = g_px[e] < wrem ? g_lane[e] : OOB;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, vo, soff, 0));
                rg[4 * e + 0] = v[0]; rg[4 * e + 1] = v[1]; rg[4 * e + 2] = v[2]; rg[4 * e + 3] = v[3];
            }
        } else {
            const float* gsrc = p.g + (size_t)((size_t)(b * p.Hg + oy) * p.Wg) * p.ldg;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = tid + 256 * e, px = idx >> 6, c = idx & 63;
                float v = 0.f;
                if (row_ok && ox0 + px < p.Wg && cg0 + c < p.Cg) v = gsrc[(size_t)(ox0 + px) * p.ldg + cg0 + c];
                rg[e] = v;
            }
        }
        // ---- X row segment: npos positions x CISL channels ----
        const int ixb = ox0 * p.stride - p.pad;
        if (x_vec) {
            const int soff = row_ok ? (int)((unsigned)((b * p.Hx + iy) * p.Wx * p.ldx) * 4u) : 0;
            const unsigned ldx4 = (unsigned)p.ldx * 4u;
#pragma unroll
            for (int e = 0; e < XV; ++e) {
                int ix = ixb + x_pos[e];
                if (p.pad_mode == 1) ix = reflect_idx(ix, p.Wx);
                const bool ok = row_ok && ix >= 0 && ix < p.Wx;     // positions past the row pair with zero G
                const unsigned vo = ok ? (unsigned)ix * ldx4 + x_lane[e] : OOB;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, soff, 0));
                rx[4 * e + 0] = v[0]; rx[4 * e + 1] = v[1]; rx[4 * e + 2] = v[2]; rx[4 * e + 3] = v[3];
            }
        } else {
            const float* xsrc = p.x + (size_t)((size_t)(b * p.Hx + (row_ok ? iy : 0)) * p.Wx) * p.ldx;
            const int n1 = npos * CISL;   // <= 72*3
            const int idx = tid;
            float v = 0.f;
            if (idx < n1 && row_ok) {
                const int pos = idx / CISL, c = idx - pos * CISL;
                int ix = ixb + pos;
                if (p.pad_mode == 1) ix = reflect_idx(ix, p.Wx);
                if (ix >= 0 && ix < p.Wx && cx0 + c < p.Cx) v = xsrc[(size_t)ix * p.ldx + cx0 + c];
            }
            rx[0] = v;
        }
    };

    auto lstore = [&]() {
        if (g_vec) {
#pragma unroll
            for (int e = 0; e < GV; ++e) {
                const int idx = tid + 256 * e;
                f32x4 v = {rg[4 * e], rg[4 * e + 1], rg[4 * e + 2], rg[4 * e + 3]};
                *reinterpret_cast<f32x4*>(&Gs[idx * 4]) = v;     // [px][64]: idx*4 == px*64 + c
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) Gs[tid + 256 * e] = rg[e];
        }
        if (x_vec) {
            const int n4 = npos * (CISL >> 2);
#pragma unroll
            for (int e = 0; e < XV; ++e) {
                const int idx = tid + 256 * e;
                if (idx < n4) {
                    f32x4 v = {rx[4 * e], rx[4 * e + 1], rx[4 * e + 2], rx[4 * e + 3]};
                    *reinterpret_cast<f32x4*>(&Xs[idx * 4]) = v;  // [pos][CISL]
                }
            }
        } else {
            if (tid < npos * CISL) Xs[tid] = rx[0];
        }
    };

    const int h = lane >> 5;
    // Straight-line inner loop: every wave runs exactly WG_NTW column tiles (wc + 2t); columns past
    // k*slab read stale-but-finite LDS words whose products are never stored (MFMA columns are
    // independent), so there is no branch and no mask between the LDS reads and the MFMAs.
    const int nsub = WG_TW / 2;      // MFMA k-substeps per segment; lane half h owns pixels [h*nsub, (h+1)*nsub)
    const float* Gp = Gs + (h * nsub) * WG_ROWS + wr * 32 + (lane & 31);
    const int xstep = p.stride * CISL;
    const float* Xp = Xs + (h * nsub) * xstep + wc * 32 + (lane & 31);

    if (total > 0) {
        gload(0);
        lstore();
        __syncthreads();
        for (int sidx = 0; sidx < total; ++sidx) {
            if (sidx + 1 < total) gload(sidx + 1);
#pragma unroll 4
            for (int s = 0; s < nsub; ++s) {
                const float a = Gp[s * WG_ROWS];
                const float* xr = Xp + s * xstep;
#pragma unroll
                for (int t = 0; t < WG_NTW; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xr[t * 64], acc[t], 0, 0, 0);
            }
            __syncthreads();
            if (sidx + 1 < total) {
                lstore();
                __syncthreads();
            }
        }
    }

    // epilogue: partial slab [split][ky*k + kx][Cg][Cx]
    const int KK = p.k * p.k;
#pragma unroll
    for (int t = 0; t < WG_NTW; ++t) {
        const int tj = wc + 2 * t;
        if (tj >= NT) continue;
        const int jj = tj * 32 + (lane & 31);
        if (jj >= ncols) continue;
        const int kx = jj / CISL, cl = jj - kx * CISL, cx = cx0 + cl;
        if (cx >= p.Cx) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cg = cg0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (cg < p.Cg)
                p.part[(((size_t)split * KK + ky * p.k + kx) * p.Cg + cg) * p.Cx + cx] = acc[t][r];
        }
    }
}

// Thin variant for the layers whose shifted tensor has 1-3 channels (the image-input 9x9 conv and the
// two 64->1 heads): all k*Cx columns of a filter row fit ONE 32-column MFMA tile, so a workgroup keeps
// the G tile in LDS and sweeps ALL k filter rows over it (tile t of column-wave wc is filter row
// ky = wc + 2t).  G -- the 272 MB tensor -- is read once instead of once per filter row.
#define TH_XROW 128       // floats per staged X row (>= (31 + 9) * 3)
__global__ __launch_bounds__(256) void conv_wgrad_thin_f32(const WgradParams p) {
    constexpr int TW = 32, NTW = 5;
    __shared__ __attribute__((aligned(16))) float smem[TW * WG_ROWS + 10 * TH_XROW + 64];
    float* Gs = smem;
    float* Xs = smem + TW * WG_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1, h = lane >> 5;
    const int cgt = blockIdx.x % p.n_cgt, split = blockIdx.x / p.n_cgt;
    const int Cx = p.Cx, k = p.k;
    const int npos = (TW - 1) + k, cg0 = cgt * WG_ROWS;

    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int R = p.B * p.Hg;
    const int r0 = split * p.rows_per_split, r1 = min(R, r0 + p.rows_per_split);
    const int nseg = (p.Wg + TW - 1) / TW;
    const int total = (r1 - r0) * nseg;
    const unsigned OOB = 0xFFFFFF00u;
    __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, (int)p.g_bytes, 0x00020000);
    unsigned g_lane[2]; int g_px[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = tid + 256 * e, c = (idx & 15) * 4;
        g_px[e] = idx >> 4;
        g_lane[e] = (cg0 + c < p.Cg) ? (unsigned)(g_px[e] * p.ldg + cg0 + c) * (p.g_bf16 ? 2u : 4u) : OOB;
    }
    // X staging slots: idx -> (filter row, position*Cx + channel)
    int x_ky[5], x_q[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        const int idx = tid + 256 * e;
        x_ky[e] = idx / TH_XROW;
        x_q[e] = idx - x_ky[e] * TH_XROW;
        if (x_ky[e] >= k || x_q[e] >= npos * Cx) x_ky[e] = -1;
    }
    f32x4 rg[2];
    float rx[5];

    auto gload = [&](int sidx) {
        const int r = r0 + sidx / nseg, ox0 = (sidx % nseg) * TW;
        const int b = r / p.Hg, oy = r % p.Hg;
        const int soff = (int)((unsigned)(((b * p.Hg + oy) * p.Wg + ox0) * p.ldg) * (p.g_bf16 ? 2u : 4u));
        const int wrem = min(p.Wg - ox0, TW);
        if (p.g_bf16) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 u = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_g, g_px[e] < wrem ? g_lane[e] : OOB, soff, 0));
                rg[e][0] = __uint_as_float(u[0] << 16); rg[e][1] = __uint_as_float(u[0] & 0xffff0000u);
                rg[e][2] = __uint_as_float(u[1] << 16); rg[e][3] = __uint_as_float(u[1] & 0xffff0000u);
            }
        } else
#pragma unroll
        for (int e = 0; e < 2; ++e)
            rg[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, g_px[e] < wrem ? g_lane[e] : OOB, soff, 0));
        const int ixb = ox0 - p.pad;
#pragma unroll
        for (int e = 0; e < 5; ++e) {
            float v = 0.f;
            if (x_ky[e] >= 0) {
                int iy = oy - p.pad + x_ky[e];
                const int pos = x_q[e] / Cx, c = x_q[e] - pos * Cx;
                int ix = ixb + pos;
                if (p.pad_mode == 1) { iy = reflect_idx(iy, p.Hx); ix = reflect_idx(ix, p.Wx); }
                if ((unsigned)iy < (unsigned)p.Hx && (unsigned)ix < (unsigned)p.Wx)
                    v = p.x[((size_t)(b * p.Hx + iy) * p.Wx + ix) * p.ldx + c];
            }
            rx[e] = v;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int e = 0; e < 2; ++e) *reinterpret_cast<f32x4*>(&Gs[(tid + 256 * e) * 4]) = rg[e];
#pragma unroll
        for (int e = 0; e < 5; ++e)
            if (tid + 256 * e < 10 * TH_XROW) Xs[tid + 256 * e] = rx[e];
    };

    const float* Gp = Gs + (h * 16) * WG_ROWS + wr * 32 + (lane & 31);
    const float* Xp = Xs + wc * TH_XROW + (h * 16) * Cx + (lane & 31);
    if (total > 0) {
        gload(0);
        lstore();
        __syncthreads();
        for (int sidx = 0; sidx < total; ++sidx) {
            if (sidx + 1 < total) gload(sidx + 1);
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const float a = Gp[s * WG_ROWS];
                const float* xr = Xp + s * Cx;
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xr[t * 2 * TH_XROW], acc[t], 0, 0, 0);
            }
            __syncthreads();
            if (sidx + 1 < total) {
                lstore();
                __syncthreads();
            }
        }
    }
    const int KK = k * k, jj = lane & 31;
    if (jj >= k * Cx) return;
    const int kx = jj / Cx, cl = jj - kx * Cx;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int ky = wc + 2 * t;
        if (ky >= k) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cg = cg0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (cg < p.Cg) p.part[(((size_t)split * KK + ky * k + kx) * p.Cg + cg) * Cx + cl] = acc[t][r];
        }
    }
}

// Sum the split-K slabs in a fixed order and scatter to the destination layout.
//   part [S][KK][R][C]  ->  dw[tap'][..]: element (tap, r, c) goes to
//   transpose == 0: dw[(tap' * R + r) * ld + off + c]
//   transpose == 1: dw[(tap' * C + c) * ld + off + r]
//   tap' = flip ? KK-1-tap : tap
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int KK, int R,
                                    int C, int ld, int off, int transpose, int flip) {
    const int64_t n = (int64_t)KK * R * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        // eight slabs per trip, loaded together (from clamped addresses) and added in slab order: the launches with few outputs
        // and hundreds of slabs (16-61 workgroups) were one dependent load at a time, 120-150 us for a few KB of gradient
        for (int sp = 0; sp < S; sp += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = part[(size_t)(sp + q < S ? sp + q : S - 1) * n + i];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (sp + q < S) s += v[q];
        }
        const int c = (int)(i % C);
        const int64_t t2 = i / C;
        const int r = (int)(t2 % R);
        int tap = (int)(t2 / R);
        if (flip) tap = KK - 1 - tap;
        if (transpose) dw[((size_t)tap * C + c) * ld + off + r] = s;
        else dw[((size_t)tap * R + r) * ld + off + c] = s;
    }
}

namespace {

struct WgradPlan {
    WgradParams P;
    int transpose, flip;
    bool thin;
    size_t ws_bytes;
    int blocks;
};

// Maps a layer's weight gradient onto the (G, X) roles of the kernel.
bool make_plan(const gdn_conv_geom* g, int Cx_in, WgradPlan& pl) {
    if (!g || g->k < 1 || g->k > 9 || g->stride < 1 || g->stride > 2) return false;
    int Ho, Wo;
    if (gdn_conv_out_dims(g, &Ho, &Wo) != GDN_OK) return false;
    WgradParams& P = pl.P;
    P = WgradParams{};
    P.B = g->B; P.k = g->k; P.stride = g->stride; P.pad = g->pad; P.pad_mode = g->pad_mode;
    pl.transpose = 0; pl.flip = 0;
    if (!g->transposed) {
        if (g->Cout < 32 && g->stride == 1 && g->pad_mode == 0) {
            // Few output channels (the 64->1 head): put the INPUT channels on the MFMA rows.
            // dW[tap][co][ci] = sum_q X[q][ci] * dY[q - tap + p][co]: X unshifted, dY shifted
            // by the flipped tap with padding k-1-p.
            P.Hg = g->H; P.Wg = g->W; P.Cg = Cx_in;          // G role: x
            P.Hx = Ho; P.Wx = Wo; P.Cx = g->Cout;            // X role: dy
            P.pad = g->k - 1 - g->pad;
            pl.transpose = 1; pl.flip = 1;
        } else {
            P.Hg = Ho; P.Wg = Wo; P.Cg = g->Cout;            // G role: dy
            P.Hx = g->H; P.Wx = g->W; P.Cx = Cx_in;          // X role: x
        }
    } else {
        // ConvTranspose2d: dW[ci][co][tap] = sum_i X[i][ci] * dY[i*s - p + tap][co]
        P.Hg = g->H; P.Wg = g->W; P.Cg = Cx_in;              // G role: x (layer input)
        P.Hx = Ho; P.Wx = Wo; P.Cx = g->Cout;                // X role: dy
        pl.transpose = 1;
    }
    pl.thin = P.Cx * P.k <= 32 && P.k <= 9 && P.stride == 1 && (P.Cg % 4 == 0) && ((P.k - 1 + 32) * P.Cx <= TH_XROW);
    if (pl.thin) {
        P.cisl = P.Cx; P.tw = 32;
        P.n_cgt = cdiv(P.Cg, WG_ROWS); P.n_cxt = 1;
        const int R = P.B * P.Hg;
        int S = cdiv(768, P.n_cgt);
        if (S > R) S = R;
        P.rows_per_split = cdiv(R, S);
        P.S = cdiv(R, P.rows_per_split);
        pl.blocks = P.n_cgt * P.S;
        pl.ws_bytes = (size_t)P.S * P.k * P.k * P.Cg * P.Cx * sizeof(float);
        return true;
    }
    P.cisl = P.Cx < WG_SLAB ? P.Cx : WG_SLAB;
    if (P.Cx > WG_SLAB && (P.Cx % WG_SLAB)) return false;
    // segment width: even, tiles the row exactly when the width allows it
    {
        const bool thin = (P.Cg % 4 != 0) || (P.cisl % 4 != 0);      // scalar-load layers keep 32
        int best = 32;
        if (!thin) {
            // 32 is the cheapest variant (compile-time trip counts, half the staging registers): another
            // width must use the row at least 3 % better to win
            double best_eff = (double)P.Wg / (double)(cdiv(P.Wg, 32) * 32);
            for (int tw = 64; tw >= 16; tw -= 2) {
                const double eff = (double)P.Wg / (double)(cdiv(P.Wg, tw) * tw) * (tw >= 32 ? 1.0 : 0.97);
                if (eff > best_eff + 0.03) { best_eff = eff; best = tw; }
            }
        }
        P.tw = best;
    }
    if (((P.k * P.cisl + 31) / 32 + 1) / 2 > WG_NTW_MAX) return false;
    if (P.cisl % 4 != 0 && ((P.tw - 1) * P.stride + P.k) * P.cisl > 256) return false;
    P.n_cgt = cdiv(P.Cg, WG_ROWS);
    P.n_cxt = cdiv(P.Cx, P.cisl);
    const int base = P.k * P.n_cgt * P.n_cxt;
    const int R = P.B * P.Hg;
    // Split-K factor: fill the chip with whole "rounds" of resident workgroups (256 CUs x occupancy
    // of the instantiation), at least one full round, at most ~6.
    {
        const int ntw = ((P.k * P.cisl + 31) / 32 + 1) / 2;
        const bool tw32 = P.tw == 32;
        const int occ = ntw <= 1 ? (tw32 ? 4 : 3) : ntw <= 3 ? (tw32 ? 3 : 2) : ntw <= 7 ? 2 : (tw32 ? 2 : 1);
        const int slots = 256 * occ;
        int bestS = 1; double best = -1.0;
        const int smax = R < 256 ? R : 256;
        for (int S = 1; S <= smax; ++S) {
            const int rps = cdiv(R, S);
            const int Se = cdiv(R, rps);                  // effective split count
            const double fill = (double)base * Se / slots;
            if (fill > 4.0 && best > 0) break;
            const double rounds = fill < 1.0 ? 1.0 : (double)cdiv(base * Se, slots);
            double eff = fill / rounds;                   // resident-slot utilisation
            eff *= (double)R / ((double)rps * Se);        // rows wasted in the last split
            eff *= 1.0 - 0.15 / rounds;                   // several rounds balance better than one (measured)
            if (eff > best + 0.01) { best = eff; bestS = S; }
        }
        P.rows_per_split = cdiv(R, bestS);
        P.S = cdiv(R, P.rows_per_split);
    }
    pl.blocks = base * P.S;
    pl.ws_bytes = (size_t)P.S * P.k * P.k * P.Cg * P.Cx * sizeof(float);
    return true;
}

}  // namespace

extern "C" size_t gdn_conv_wgrad_workspace_bytes(const gdn_conv_geom* g, int32_t Cx) {
    WgradPlan pl;
    if (!make_plan(g, Cx, pl)) return 0;
    return pl.ws_bytes;
}

extern "C" int gdn_conv_wgrad(const gdn_conv_geom* g, const void* xv, int32_t ldx, int32_t Cx, const void* dyv,
                              int32_t ldy, float* dw, int32_t ld_dw, int32_t ci_off, void* workspace,
                              size_t workspace_bytes, int32_t dtypes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    const float *x = (const float*)xv, *dy = (const float*)dyv;
    if (!x || !dy || !dw) return GDN_ERR_BAD_ARG;
    WgradPlan pl;
    if (!make_plan(g, Cx, pl)) return GDN_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < pl.ws_bytes) return GDN_ERR_WORKSPACE;
    WgradParams& P = pl.P;
    const bool x_is_g = pl.transpose != 0;   // roles swapped: layer input is the G tensor
    if (x_is_g) { P.g = x; P.ldg = ldx; P.x = dy; P.ldx = ldy; }
    else { P.g = dy; P.ldg = ldy; P.x = x; P.ldx = ldx; }
    // mixed precision: only the wide (>= 4 channel) tensor of a thin layer may hold bf16
    if (dtypes) {
        const int g_bit = x_is_g ? 1 : 2;
        if (!pl.thin || dtypes != g_bit || (P.ldg % 4)) return GDN_ERR_UNSUPPORTED;
        P.g_bf16 = 1;
    }
    P.part = (float*)workspace;
    {
        const uint64_t gb = (((uint64_t)P.B * P.Hg * P.Wg - 1) * (uint64_t)P.ldg + P.Cg) * (P.g_bf16 ? 2 : 4);
        const uint64_t xb = (((uint64_t)P.B * P.Hx * P.Wx - 1) * (uint64_t)P.ldx + P.Cx) * 4;
        if (gb >= 0xFF000000ull || xb >= 0xFF000000ull) return GDN_ERR_UNSUPPORTED;
        P.g_bytes = (unsigned)gb; P.x_bytes = (unsigned)xb;
    }
    hipStream_t st = (hipStream_t)stream;
    const int ntw = ((P.k * P.cisl + 31) / 32 + 1) / 2;   // column tiles per wave
    const dim3 grid(pl.blocks), blk(256);
    if (pl.thin) {
        if (P.ldg % 4) return GDN_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(conv_wgrad_thin_f32, grid, blk, 0, st, P);
    } else
#define WG_LAUNCH(N)                                                                          \
    do {                                                                                      \
        /* narrow rows (tw < 32, level 4) also fit the 32-pixel instantiation: half the LDS, one more wave per SIMD */  \
        if (P.tw == 32 || (P.tw < 32 && N <= 5)) hipLaunchKernelGGL((conv_wgrad_f32<N, 32>), grid, blk, 0, st, P);     \
        else hipLaunchKernelGGL((conv_wgrad_f32<N, 64>), grid, blk, 0, st, P);                \
    } while (0)
    if (ntw <= 1) WG_LAUNCH(1);
    else if (ntw <= 3) WG_LAUNCH(3);
    else if (ntw <= 5) WG_LAUNCH(5);
    else if (ntw <= 7) WG_LAUNCH(7);
    else WG_LAUNCH(9);
#undef WG_LAUNCH
    int rc = gdn_launch_status();
    if (rc != GDN_OK) return rc;
    const int KK = P.k * P.k;
    const int64_t n = (int64_t)KK * P.Cg * P.Cx;
    const int blocks = (int)(cdiv64(n, 256) < 2048 ? cdiv64(n, 256) : 2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, P.S, KK, P.Cg,
                       P.Cx, ld_dw, ci_off, pl.transpose, pl.flip);
    return gdn_launch_status();
}
