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

#define WG_TW 32          // pixels per row segment (MFMA K' per LDS image)
#define WG_ROWS 64        // channels of G per workgroup
#define WG_SLAB 64        // max channels of X per workgroup
#define WG_MAXPOS 72      // >= 31*2 + 9
#define WG_NTW_MAX 9      // max 32-column tiles per wave

struct WgradParams {
    const float* g; const float* x; float* part;
    int B, Hg, Wg, ldg, Cg;
    int Hx, Wx, ldx, Cx;
    int k, stride, pad, pad_mode;
    int cisl, n_cgt, n_cxt, S, rows_per_split;
};

template <int WG_NTW>
__global__ __launch_bounds__(256) void conv_wgrad_f32(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) float smem[WG_TW * WG_ROWS + WG_MAXPOS * WG_SLAB + 64];
    float* Gs = smem;
    float* Xs = smem + WG_TW * WG_ROWS;

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

    float rg[8], rx[20];

    auto gload = [&](int sidx) {
        const int r = r0 + sidx / nseg, ox0 = (sidx % nseg) * WG_TW;
        const int b = r / p.Hg, oy = r % p.Hg;
        int iy = oy * p.stride - p.pad + ky;
        bool row_ok = true;
        if (p.pad_mode == 1) iy = reflect_idx(iy, p.Hx);
        row_ok = iy >= 0 && iy < p.Hx;
        // ---- G tile: 32 pixels x 64 channels ----
        const float* gsrc = p.g + (size_t)((size_t)(b * p.Hg + oy) * p.Wg) * p.ldg;
        if (g_vec) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = tid + 256 * e, px = idx >> 4, c = (idx & 15) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row_ok && ox0 + px < p.Wg && cg0 + c < p.Cg)
                    v = *reinterpret_cast<const f32x4*>(gsrc + (size_t)(ox0 + px) * p.ldg + cg0 + c);
                rg[4 * e + 0] = v[0]; rg[4 * e + 1] = v[1]; rg[4 * e + 2] = v[2]; rg[4 * e + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = tid + 256 * e, px = idx >> 6, c = idx & 63;
                float v = 0.f;
                if (row_ok && ox0 + px < p.Wg && cg0 + c < p.Cg) v = gsrc[(size_t)(ox0 + px) * p.ldg + cg0 + c];
                rg[e] = v;
            }
        }
        // ---- X row segment: npos positions x CISL channels ----
        const float* xsrc = p.x + (size_t)((size_t)(b * p.Hx + iy) * p.Wx) * p.ldx;
        const int ixb = ox0 * p.stride - p.pad;
        if (x_vec) {
            const int q4 = CISL >> 2, n4 = npos * q4;
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                const int idx = tid + 256 * e;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (idx < n4 && row_ok) {
                    const int pos = idx / q4, c = (idx - pos * q4) * 4;
                    int ix = ixb + pos;
                    bool ok = true;
                    if (p.pad_mode == 1) ix = reflect_idx(ix, p.Wx);
                    ok = ix >= 0 && ix < p.Wx;     // positions past the row's last pixel pair with zero G
                    if (ok && cx0 + c < p.Cx) v = *reinterpret_cast<const f32x4*>(xsrc + (size_t)ix * p.ldx + cx0 + c);
                }
                rx[4 * e + 0] = v[0]; rx[4 * e + 1] = v[1]; rx[4 * e + 2] = v[2]; rx[4 * e + 3] = v[3];
            }
        } else {
            const int n1 = npos * CISL;   // <= 72*3
#pragma unroll
            for (int e = 0; e < 1; ++e) {
                const int idx = tid + 256 * e;
                float v = 0.f;
                if (idx < n1 && row_ok) {
                    const int pos = idx / CISL, c = idx - pos * CISL;
                    int ix = ixb + pos;
                    bool ok = true;
                    if (p.pad_mode == 1) ix = reflect_idx(ix, p.Wx);
                    ok = ix >= 0 && ix < p.Wx;
                    if (ok && cx0 + c < p.Cx) v = xsrc[(size_t)ix * p.ldx + cx0 + c];
                }
                rx[e] = v;
            }
        }
    };

    auto lstore = [&]() {
        if (g_vec) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
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
            for (int e = 0; e < 5; ++e) {
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
    const float* Gp = Gs + (h * (WG_TW / 2)) * WG_ROWS + wr * 32 + (lane & 31);
    const int xstep = p.stride * CISL;
    const float* Xp = Xs + (h * (WG_TW / 2)) * xstep + wc * 32 + (lane & 31);

    if (total > 0) {
        gload(0);
        lstore();
        __syncthreads();
        for (int sidx = 0; sidx < total; ++sidx) {
            if (sidx + 1 < total) gload(sidx + 1);
#pragma unroll 4
            for (int s = 0; s < WG_TW / 2; ++s) {
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
        for (int sp = 0; sp < S; ++sp) s += part[(size_t)sp * n + i];
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
    P.cisl = P.Cx < WG_SLAB ? P.Cx : WG_SLAB;
    if (P.Cx > WG_SLAB && (P.Cx % WG_SLAB)) return false;
    if ((WG_TW - 1) * P.stride + P.k > WG_MAXPOS) return false;
    if (((P.k * P.cisl + 31) / 32 + 1) / 2 > WG_NTW_MAX) return false;
    if (P.cisl % 4 != 0 && ((WG_TW - 1) * P.stride + P.k) * P.cisl > 256) return false;
    P.n_cgt = cdiv(P.Cg, WG_ROWS);
    P.n_cxt = cdiv(P.Cx, P.cisl);
    const int base = P.k * P.n_cgt * P.n_cxt;
    const int R = P.B * P.Hg;
    int S = cdiv(1024, base);
    if (S < 1) S = 1;
    if (S > R) S = R;
    P.rows_per_split = cdiv(R, S);
    P.S = cdiv(R, P.rows_per_split);
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

extern "C" int gdn_conv_wgrad(const gdn_conv_geom* g, const float* x, int32_t ldx, int32_t Cx, const float* dy,
                              int32_t ldy, float* dw, int32_t ld_dw, int32_t ci_off, void* workspace,
                              size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!x || !dy || !dw) return GDN_ERR_BAD_ARG;
    WgradPlan pl;
    if (!make_plan(g, Cx, pl)) return GDN_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < pl.ws_bytes) return GDN_ERR_WORKSPACE;
    WgradParams& P = pl.P;
    const bool x_is_g = pl.transpose != 0;   // roles swapped: layer input is the G tensor
    if (x_is_g) { P.g = x; P.ldg = ldx; P.x = dy; P.ldx = ldy; }
    else { P.g = dy; P.ldg = ldy; P.x = x; P.ldx = ldx; }
    P.part = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int ntw = ((P.k * P.cisl + 31) / 32 + 1) / 2;   // column tiles per wave
    if (ntw <= 1) hipLaunchKernelGGL(conv_wgrad_f32<1>, dim3(pl.blocks), dim3(256), 0, st, P);
    else if (ntw <= 3) hipLaunchKernelGGL(conv_wgrad_f32<3>, dim3(pl.blocks), dim3(256), 0, st, P);
    else if (ntw <= 5) hipLaunchKernelGGL(conv_wgrad_f32<5>, dim3(pl.blocks), dim3(256), 0, st, P);
    else if (ntw <= 7) hipLaunchKernelGGL(conv_wgrad_f32<7>, dim3(pl.blocks), dim3(256), 0, st, P);
    else hipLaunchKernelGGL(conv_wgrad_f32<9>, dim3(pl.blocks), dim3(256), 0, st, P);
    int rc = gdn_launch_status();
    if (rc != GDN_OK) return rc;
    const int KK = P.k * P.k;
    const int64_t n = (int64_t)KK * P.Cg * P.Cx;
    const int blocks = (int)(cdiv64(n, 256) < 2048 ? cdiv64(n, 256) : 2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, P.S, KK, P.Cg,
                       P.Cx, ld_dw, ci_off, pl.transpose, pl.flip);
    return gdn_launch_status();
}
