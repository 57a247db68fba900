// FFT-domain convolution for the large-window residual layers (9x9 on 64 channels, 7x7 on 128, 5x5 on 256): 80 % of G's FLOPs.
//
// A k x k stride-1 "same" convolution costs k*k MACs per (pixel, cin, cout); in the frequency domain it costs one
// complex MAC (4 real) per (frequency bin, cin, cout) plus transforms that are linear in the tensor size.  With 32x32
// tiles (overlap-save: T = 33 - k valid outputs per tile side, T = 24 for 9x9, 26 for 7x7) and the 17 x 32 bins a real
// input needs, the multiply count per output drops 81 -> 544*4/24^2 = 3.8 (9x9) and 49 -> 3.2 (7x7), and -- at fp32
// MFMA speed, where the direct kernels already sit at 0.7-0.8 of peak -- that is the only lever left that is worth a
// factor.  Accuracy is not traded: each output sums 64-128 products per bin instead of 5184-6272, and measured against
// an fp64 convolution the tiled fp32 FFT result is as close as the direct fp32 sum (DESIGN.md 2.4,
// tests/test_fftconv_model_cpu.py).
//
// Pipeline (all tensors NHWC, channels contiguous; spectra are [bin = ky*17+kx][tile][channel] complex):
//   fft2d_fwd    x  -> Xf     real FFT32 along x then FFT32 along y of each 32x32 patch (zero padding = halo); one
//                             workgroup = one tile x 16 channels, the two passes meet in LDS
//   weights      w  -> Wf     [bin][3][Cout][Cin] real: conj(DFT) of the k*k taps (correlation) as the three planes of the
//                             3-multiplication complex product (Re, Im - Re, Re + Im)
//   cgemm_bins   Yf[bin] = Xf[bin] * Wf[bin]^T      one complex GEMM per bin, M = tiles, three real MFMA products
//   ifft2d_valid Yf -> y      inverse FFT32 along ky, Hermitian inverse along kx, valid T x T outputs, 1/1024 scale,
//                             affine / ReLU / residual epilogue, BatchNorm sum / sum-of-squares partials (slot = tile)
// Backward from ONE transform of dy (tile without halo, zero padded = the linear convolution fits the 32-point circle):
//   data gradient    Ef[bin] = Df[bin] * conj(Wf[bin]) (the same saved planes, read row-wise), inverse along ky, then overlap-add of the
//                    32x32 patches at offset -pad: vertical overlaps summed in the frequency domain, horizontal ones by two
//                    ordered launches (even tiles store, odd tiles add) -- deterministic, no atomics
//   weight gradient  P[bin] = Df[bin]^T * Xf[bin] (reduction over tiles, MFMA), inverse DFT at the k*k taps only
#include "common.h"
#include <cstdlib>
#include "up2x.h"

// Tile size NP (points per side): 32 for the 7x7 / 9x9 layers, 16 for the 5x5 (and 3x3) layers on >= 256 channels, whose weight
// spectrum would otherwise outweigh the activations' (544 bins x 256 x 256 x 3 planes = 428 MB per layer; 113 MB at 144 bins)
// and whose 32 x 104 images tile badly with T = 28 (56 x 112 computed for 32 x 104).  NK = NP/2 + 1 kx bins are kept of a
// real row transform; bins = NP * NK is a multiple of 8 for both sizes (they are dealt to the 8 XCDs).
#define FFT_NK_OF(NP) ((NP) / 2 + 1)
#define FFT_BINS_OF(NP) ((NP) * FFT_NK_OF(NP))
static_assert(FFT_BINS_OF(32) % 8 == 0 && FFT_BINS_OF(16) % 8 == 0 && FFT_BINS_OF(40) % 8 == 0, "bins are dealt to the 8 XCDs");
// Index arithmetic is kept off the vector ALU (the transform kernels are VALU-bound; 64-bit divisions and per-element
// 64-bit multiplies were most of their instructions): grids carry (channel chunk, tile), row and image instead of a
// flat index, and every strided access walks a running pointer.  GDN_KEEP pins a running value so the unrolled loops do
// not fold it back into multiplies.
#define GDN_KEEP(v) asm volatile("" : "+v"(v))

// Round-6 A/B switches (tools/ab_variant.sh ... -DGDN_x=n; measured in profiles/r06_fft_ab.txt):
#ifndef GDN_ICOLS_ORDER
#define GDN_ICOLS_ORDER 1    // ifft_cols grid: 0: blockIdx.x = tile group, y = kx;  1: blockIdx.x = kx (fastest), y = tile group
#endif
#ifndef GDN_FFT_DGRAD_PATCH
#define GDN_FFT_DGRAD_PATCH 1  // data gradient's inverse: 1 = single-pass inverse into a patch buffer + gather (overlap-add), 0 = ifft_cols + two
#endif                         //                           ordered ifft_rows_overlap launches through the intermediate S (rounds 1-5)
#ifndef GDN_ROWS_REMAP
#define GDN_ROWS_REMAP 2     // ifft_rows_overlap: 0 = 4 tiles x 64-channel chunk, 1 = (256 / C) tiles x all channels, 2 = 1 only for C = 256
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));

namespace {

// Raw buffer access for the transform kernels (round 6).  Walking a global pointer through an `asm volatile` pin (GDN_KEEP) made
// the compiler forget the address space: every strided access became a FLAT instruction with two vector instructions of 64-bit
// address arithmetic in front of it.  Here the part of an address that is uniform over the workgroup (tile, bin row, column) lives
// in the descriptor's base -- scalar arithmetic -- and the per-lane part is a 32-bit byte offset that is a constant of the thread.
// The descriptor spans 4 GiB from its base; what is in range is decided by the caller (a per-lane offset >= 2^31 reads as zero).
// Values that are uniform over the workgroup but were computed on the vector unit (an integer division of blockIdx) live in vector
// registers: a descriptor built from them makes the compiler emit a "waterfall" loop around every access.  GDN_UNI moves them to
// scalar registers once, where they are defined.
#define GDN_UNI(v) __builtin_amdgcn_readfirstlane(v)
template <class T>
__device__ __forceinline__ T* fft_uni_ptr(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = GDN_UNI((unsigned)v), hi = GDN_UNI((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fft_rs(const void* base, unsigned bytes = 0xffffffffu) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x2 fft_ld2(const void* base, unsigned voff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(fft_rs(base), voff, 0, 0));
}
__device__ __forceinline__ void fft_st2(void* base, unsigned voff, float a, float b) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2_t, f32x2{a, b}), fft_rs(base), voff, 0, 0);
}
__device__ __forceinline__ float fft_ld1(const void* base, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fft_rs(base), voff, 0, 0));
}
__device__ __forceinline__ void fft_st1(void* base, unsigned voff, float a) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a), fft_rs(base), voff, 0, 0);
}
constexpr unsigned FFT_VOFF_ZERO = 0x80000000u;      // a per-lane offset no descriptor reaches: the load returns 0, the store is dropped

// cos/sin(2*pi*j/32), j = 0..31
__device__ __constant__ float kCos32[32] = {
    1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f, 0.55557023301960229f,
    0.38268343236508984f, 0.19509032201612833f, 0.0f, -0.19509032201612819f, -0.38268343236508973f, -0.55557023301960196f,
    -0.70710678118654746f, -0.83146961230254535f, -0.92387953251128674f, -0.98078528040323043f, -1.0f,
    -0.98078528040323043f, -0.92387953251128685f, -0.83146961230254546f, -0.70710678118654768f, -0.55557023301960218f,
    -0.38268343236509034f, -0.19509032201612866f, 0.0f, 0.19509032201612830f, 0.38268343236509000f, 0.55557023301960184f,
    0.70710678118654735f, 0.83146961230254524f, 0.92387953251128652f, 0.98078528040323032f};
__device__ __constant__ float kSin32[32] = {
    0.0f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f, 0.70710678118654746f, 0.83146961230254524f,
    0.92387953251128674f, 0.98078528040323043f, 1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254546f,
    0.70710678118654757f, 0.55557023301960218f, 0.38268343236508989f, 0.19509032201612861f, 0.0f,
    -0.19509032201612836f, -0.38268343236508967f, -0.55557023301960196f, -0.70710678118654746f, -0.83146961230254524f,
    -0.92387953251128652f, -0.98078528040323032f, -1.0f, -0.98078528040323043f, -0.92387953251128663f,
    -0.83146961230254546f, -0.70710678118654768f, -0.55557023301960218f, -0.38268343236509039f, -0.19509032201612872f};

// In-register radix-2 decimation-in-time FFT of NP (32 or 16) complex points.  SIGN = -1 forward, +1 inverse (unscaled).
// The loops are fully unrolled and the twiddles are compile-time literals, so the trivial ones (1, -+i) cost no multiplies
// and the (1 -+ i)/sqrt2 ones two.
template <int NP, int SIGN>
__device__ __forceinline__ void fft_pow2(float (&re)[NP], float (&im)[NP]) {
    static_assert(NP == 32 || NP == 16 || NP == 8, "power-of-two transform");
    constexpr int LOG = NP == 32 ? 5 : NP == 16 ? 4 : 3, SC = 32 / NP;
    constexpr float C32[9] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f,
                              0.55557023301960229f, 0.38268343236508984f, 0.19509032201612833f, 0.0f};
    // bit reversal (LOG bits): pure register renaming after unrolling
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        int j = 0;
#pragma unroll
        for (int b = 0; b < LOG; ++b) j |= ((i >> b) & 1) << (LOG - 1 - b);
        if (j > i) {
            const float tr = re[i], ti = im[i];
            re[i] = re[j]; im[i] = im[j];
            re[j] = tr; im[j] = ti;
        }
    }
#pragma unroll
    for (int s = 1; s <= LOG; ++s) {
        const int half = 1 << (s - 1), span = 1 << s, tstep = NP >> s;
#pragma unroll
        for (int k = 0; k < NP; k += span) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const int q = j * tstep * SC;            // twiddle angle 2*pi*q/32, q in [0, 16)
                const int a = k + j, b = a + half;
                float xr, xi;
                if (q == 0) {
                    xr = re[b]; xi = im[b];
                } else if (q == 8) {                     // w = SIGN * i
                    xr = -SIGN * im[b]; xi = SIGN * re[b];
                } else if (q == 4) {                     // (1 + SIGN i) / sqrt2
                    xr = (re[b] - SIGN * im[b]) * C32[4]; xi = (im[b] + SIGN * re[b]) * C32[4];
                } else if (q == 12) {                    // (-1 + SIGN i) / sqrt2
                    xr = (-re[b] - SIGN * im[b]) * C32[4]; xi = (SIGN * re[b] - im[b]) * C32[4];
                } else {
                    const float wr = q < 8 ? C32[q] : -C32[16 - q];
                    const float wi = SIGN * (q < 8 ? C32[8 - q] : C32[q - 8]);
                    xr = re[b] * wr - im[b] * wi; xi = re[b] * wi + im[b] * wr;
                }
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
        }
    }
}

// cos/sin(2*pi*j/40), j = 0..39 (40-point tiles)
__device__ __constant__ float kCos40[40] = {
    1.0f, 0.98768834059513777f, 0.95105651629515353f, 0.89100652418836790f, 0.80901699437494745f, 0.70710678118654757f,
    0.58778525229247314f, 0.45399049973954680f, 0.30901699437494745f, 0.15643446504023092f, 0.0f, -0.15643446504023081f,
    -0.30901699437494734f, -0.45399049973954669f, -0.58778525229247303f, -0.70710678118654746f, -0.80901699437494734f,
    -0.89100652418836779f, -0.95105651629515353f, -0.98768834059513766f, -1.0f, -0.98768834059513777f, -0.95105651629515364f,
    -0.89100652418836812f, -0.80901699437494756f, -0.70710678118654768f, -0.58778525229247325f, -0.45399049973954692f,
    -0.30901699437494756f, -0.15643446504023104f, 0.0f, 0.15643446504023067f, 0.30901699437494723f, 0.45399049973954664f,
    0.58778525229247292f, 0.70710678118654735f, 0.80901699437494734f, 0.89100652418836779f, 0.95105651629515353f,
    0.98768834059513766f};
__device__ __constant__ float kSin40[40] = {
    0.0f, 0.15643446504023087f, 0.30901699437494740f, 0.45399049973954675f, 0.58778525229247314f, 0.70710678118654746f,
    0.80901699437494745f, 0.89100652418836790f, 0.95105651629515353f, 0.98768834059513777f, 1.0f, 0.98768834059513777f,
    0.95105651629515364f, 0.89100652418836790f, 0.80901699437494745f, 0.70710678118654757f, 0.58778525229247325f,
    0.45399049973954686f, 0.30901699437494751f, 0.15643446504023098f, 0.0f, -0.15643446504023073f, -0.30901699437494728f,
    -0.45399049973954625f, -0.58778525229247303f, -0.70710678118654746f, -0.80901699437494734f, -0.89100652418836779f,
    -0.95105651629515353f, -0.98768834059513766f, -1.0f, -0.98768834059513777f, -0.95105651629515364f, -0.89100652418836812f,
    -0.80901699437494756f, -0.70710678118654768f, -0.58778525229247336f, -0.45399049973954697f, -0.30901699437494762f,
    -0.15643446504023112f};

// 40-point transform = 5 x 8 Cooley-Tukey: five 8-point transforms over x[5 n1 + n2], twiddles W40^(n2 k1), then eight
// 5-point transforms (the symmetric form: two cosine and two sine combinations) to X[k1 + 8 k2].  40-point tiles carry 32
// valid outputs of a 9x9 layer: 128 x 416 is exactly 4 x 13 of them (26 % fewer points than 6 x 18 tiles of 32).
template <int SIGN>
__device__ __forceinline__ void fft40(float (&re)[40], float (&im)[40]) {
    constexpr float CS40[11] = {1.0f, 0.98768834059513777f, 0.95105651629515353f, 0.89100652418836790f, 0.80901699437494745f,
                                0.70710678118654757f, 0.58778525229247314f, 0.45399049973954680f, 0.30901699437494745f,
                                0.15643446504023092f, 0.0f};        // cos(2 pi j / 40), j = 0..10 (sin = the mirror)
    float yr[5][8], yi[5][8];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) { yr[n2][n1] = re[5 * n1 + n2]; yi[n2][n1] = im[5 * n1 + n2]; }
        fft_pow2<8, SIGN>(yr[n2], yi[n2]);
        if (n2 > 0) {
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) {
                const int q = n2 * k1;                      // angle 2 pi q / 40, q <= 28
                // cos / sin by quadrant from the first-quadrant table
                const int qq = q % 40;
                const float c = qq <= 10 ? CS40[qq] : qq <= 20 ? -CS40[20 - qq] : qq <= 30 ? -CS40[qq - 20] : CS40[40 - qq];
                const float sn = qq <= 10 ? CS40[10 - qq] : qq <= 20 ? CS40[qq - 10] : qq <= 30 ? -CS40[30 - qq] : -CS40[qq - 30];
                const float wr = c, wi = SIGN * sn;
                const float tr = yr[n2][k1] * wr - yi[n2][k1] * wi, ti = yr[n2][k1] * wi + yi[n2][k1] * wr;
                yr[n2][k1] = tr; yi[n2][k1] = ti;
            }
        }
    }
    constexpr float C1 = 0.30901699437494745f, C2 = -0.80901699437494745f;     // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float S1 = 0.95105651629515353f, S2 = 0.58778525229247314f;      // sin(2 pi / 5), sin(4 pi / 5)
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
        const float t1r = yr[1][k1] + yr[4][k1], t1i = yi[1][k1] + yi[4][k1];
        const float t2r = yr[2][k1] + yr[3][k1], t2i = yi[2][k1] + yi[3][k1];
        const float t3r = yr[1][k1] - yr[4][k1], t3i = yi[1][k1] - yi[4][k1];
        const float t4r = yr[2][k1] - yr[3][k1], t4i = yi[2][k1] - yi[3][k1];
        const float a1r = yr[0][k1] + C1 * t1r + C2 * t2r, a1i = yi[0][k1] + C1 * t1i + C2 * t2i;
        const float a2r = yr[0][k1] + C2 * t1r + C1 * t2r, a2i = yi[0][k1] + C2 * t1i + C1 * t2i;
        const float b1r = S1 * t3r + S2 * t4r, b1i = S1 * t3i + S2 * t4i;
        const float b2r = S2 * t3r - S1 * t4r, b2i = S2 * t3i - S1 * t4i;
        re[k1] = yr[0][k1] + t1r + t2r;        im[k1] = yi[0][k1] + t1i + t2i;
        // X[k1 + 8 k2] = a +- SIGN * i * b,  i * (br + i bi) = -bi + i br
        re[k1 + 8] = a1r - SIGN * b1i;         im[k1 + 8] = a1i + SIGN * b1r;
        re[k1 + 32] = a1r + SIGN * b1i;        im[k1 + 32] = a1i - SIGN * b1r;
        re[k1 + 16] = a2r - SIGN * b2i;        im[k1 + 16] = a2i + SIGN * b2r;
        re[k1 + 24] = a2r + SIGN * b2i;        im[k1 + 24] = a2i - SIGN * b2r;
    }
}

template <int NP, int SIGN>
__device__ __forceinline__ void fftn(float (&re)[NP], float (&im)[NP]) {
    if constexpr (NP == 40) fft40<SIGN>(re, im);
    else fft_pow2<NP, SIGN>(re, im);
}

// twiddle cos / sin(2 pi idx / NP) for a non-negative index product (wave-uniform: scalar loads)
template <int NP>
__device__ __forceinline__ float tw_cos(int idx) { return NP == 40 ? kCos40[idx % 40] : kCos32[(idx & (NP - 1)) * (32 / (NP == 40 ? 32 : NP))]; }
template <int NP>
__device__ __forceinline__ float tw_sin(int idx) { return NP == 40 ? kSin40[idx % 40] : kSin32[(idx & (NP - 1)) * (32 / (NP == 40 ? 32 : NP))]; }
// ... for an index already in [0, NP)
template <int NP>
__device__ __forceinline__ float tw_cos_r(int idx) { return NP == 40 ? kCos40[idx] : kCos32[idx * (32 / (NP == 40 ? 32 : NP))]; }
template <int NP>
__device__ __forceinline__ float tw_sin_r(int idx) { return NP == 40 ? kSin40[idx] : kSin32[idx * (32 / (NP == 40 ? 32 : NP))]; }

// pitch (complex elements) of one tile of the column-inverse intermediate S[t][u][kx][c]
__host__ __device__ inline size_t fft_s_pitch(int np, int C) { return (size_t)np * (np / 2 + 1) * C; }

// Layout of the data gradient's patch buffer, in floats: a patch row of np pixels x C channels is followed by 64 floats (256 bytes)
// of padding and a tile of np rows by 64 more.  Dense, every stride is a power of two times np (16 KB rows, 512 KB tiles on the
// 7x7 / 128-channel layers) and all workgroups of the chip, which walk their columns in step, write into the same few memory
// channels: ifft2d_patch<32> ran at 1.5 TB/s (351 us against 94 us of ifft2d_valid<32> for 1.5 x the bytes).  With the padding
// both strides are odd multiples of 256 bytes.  Fits the region the intermediate S occupies (32 (np + 1) <= np C).
__host__ __device__ inline size_t fft_patch_row(int np, int C) { return (size_t)np * C + 64; }
__host__ __device__ inline size_t fft_patch_tile(int np, int C) { return (size_t)np * fft_patch_row(np, C) + 64; }

struct FftGeom {
    int np, bins;                // tile size (32 / 16), kept bins = np * (np/2 + 1)
    int B, H, W, C, N;           // input [B,H,W,C], output channels N
    int k, pad, T, tiles_y, tiles_x, M;      // M = B * tiles_y * tiles_x
    int reflect;                             // input border: 0 zeros, 1 reflection (ReflectionPad2d(pad) + conv)
    int flip;                                // stride-1 ConvTranspose2d: correlation with the flipped taps
};

// weights: DFT of the k*k taps at every kept bin.  With Wc = conj(DFT(w[n][c])) (correlation) the three real planes of
// the 3-multiplication complex product are stored:  Wf[bin][0][n][c] = Re Wc,  [1] = Im Wc - Re Wc,  [2] = Re Wc + Im Wc.
//   forward        y = x * Wc        :  k1 = P0 (xr + xi), k2 = P1 xr, k3 = P2 xi;   yr = k1 - k3, yi = k1 + k2
//   data gradient  e = d * conj(Wc)  :  m1 = P0 (dr + di), m2 = P1 di, m3 = P2 dr;   er = m1 + m2, ei = m1 - m3
// so both directions read the same buffer and the weights are transformed once per step.
// thread = (n, c) with its K*K taps in registers; block = one ky; column transform first, then the 17 kx bins.
template <int K, int NP>
__global__ __launch_bounds__(256) void fft_weights_kernel(const float* __restrict__ w /* [k*k][N][C] */, float* __restrict__ Wf,
                                                          int N, int C, int flip) {
    constexpr int NK = FFT_NK_OF(NP);
    const int ky = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int c = i % C, n = i / C;
    float ur[K], ui[K];
#pragma unroll
    for (int tx = 0; tx < K; ++tx) { ur[tx] = 0.f; ui[tx] = 0.f; }
#pragma unroll
    for (int ty = 0; ty < K; ++ty) {
        const float cs = tw_cos<NP>(ky * ty), sn = tw_sin<NP>(ky * ty);
#pragma unroll
        for (int tx = 0; tx < K; ++tx) {
            const int tap = flip ? (K - 1 - ty) * K + (K - 1 - tx) : ty * K + tx;
            const float v = w[((size_t)tap * N + n) * C + c];
            ur[tx] += v * cs;
            ui[tx] -= v * sn;              // e^{-i phi}
        }
    }
    const size_t plane = (size_t)N * C;
#pragma unroll
    for (int kx = 0; kx < NK; ++kx) {
        float wr = 0.f, wi = 0.f;
#pragma unroll
        for (int tx = 0; tx < K; ++tx) {
            const float cs = tw_cos<NP>(kx * tx), sn = tw_sin<NP>(kx * tx);
            wr += ur[tx] * cs + ui[tx] * sn;       // (ur + i ui)(cs - i sn)
            wi += ui[tx] * cs - ur[tx] * sn;
        }
        const float wci = -wi;                     // conjugate
        float* dst = Wf + (size_t)(ky * NK + kx) * 3 * plane + (size_t)n * C + c;
        dst[0] = wr;
        dst[plane] = wci - wr;
        dst[2 * plane] = wr + wci;
    }
}

// Per-bin complex GEMM with three real multiplications per complex product (Gauss), fp32 MFMA:
//   DGRAD = false:  Y[bin][m][n] = sum_c X[bin][m][c] * Wc[bin][n][c]          (planes read as [n][c], c contiguous)
//   DGRAD = true :  E[bin][m][c] = sum_n D[bin][m][n] * conj(Wc[bin][n][c])    (same planes, read as [n][c] rows, c = output)
// A (spectrum of x or dy) is [bin][M][Kc] complex, interleaved.  Workgroup tile 64 tiles x 64 complex outputs, 4 waves of
// 32 x 32, three accumulator tiles each; 16 complex reduction channels per step.  The A image keeps conv_igemm_f32's
// pitch-36 layout (lane half h owns channels 8h..8h+7 of the slab, one b128 read = two complex values); the [n][c] planes use
// pitch 20 (b128 = four channels of one plane), the data-gradient planes are read row-wise (b32, conflict-free).
// Against the 4-multiplication real embedding this is 25 % fewer MFMAs; the rounding cost (k1 - k3 cancels) is 1.4x on
// the product stage, still below a direct fp32 convolution (tests/test_fftconv_model_cpu.py).
// Workgroups are dealt XCD-aware: XCD j owns the bins = j (mod 8) and walks them bin-major with the N-tiles of one M-tile
// back to back, so a bin's weight planes stay in that XCD's L2 and an A tile is fetched from the fabric once.
// Round 5: PERSISTENT workgroups, all addressing on the scalar unit.  A one-shot workgroup loads, computes and stores in turn,
// and the four workgroups of a CU do so in phase (PMC: traffic = algorithmic, matrix pipe 57 % busy on the 64-channel layers,
// whose MFMA time and HBM time are equal -- the two were adding up); and an fp32 MFMA does not hide vector-ALU work the way a
// bf16 one does (time per MFMA ~ 64 + 4 x VALU instructions per MFMA): with four k-steps per unit, the unit's
// index arithmetic, 64-bit address math and predicated epilogue came to 3.7 vector instructions per MFMA.  Now
//   * 4 x CUs workgroups walk the (bin, M-tile, N-tile) units u, u + G, ... (mixed-radix increments, no division): the first
//     slab of the NEXT unit is requested before the last k-step of the current one and written to LDS behind its last barrier,
//     and the current unit's stores drain under the next unit's MFMAs;
//   * every global access is a raw buffer access through a per-unit descriptor (the bin's slice of A / C from the unit's first
//     row on, the bin's planes of W; 12 MB at most, so spectra beyond 4 GiB work): the per-lane offset is a constant of the
//     thread, what changes inside a unit (k-slab, N-tile) rides in the scalar offset, and rows >= M fall outside the descriptor
//     (the hardware checks the per-lane offset) -- they load as zero and their stores are dropped, no predicate.
template <bool DGRAD>
__global__ __launch_bounds__(256, 4) void cgemm_bins_kernel(const float* __restrict__ A, const float* __restrict__ Wf,
                                                         float* __restrict__ Cm, int M, int Nc /* complex outputs */,
                                                         int Kc /* complex reduction */, int nunits) {
    constexpr int LDA = 36;                                 // plane images: pitch 20 ([n][c]) or 64 ([n][c] rows, DGRAD)
    __shared__ __attribute__((aligned(16))) float As[64 * LDA], Bs[3 * (DGRAD ? 16 * 64 : 64 * 20)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int NT = Nc / 64, MT = (M + 63) / 64;
    // planes: forward [p][n = output][c = reduction] (row length Kc); data gradient [p][n = reduction][c = output] (row length Nc)
    const unsigned plane_b = (unsigned)Nc * Kc * 4;         // bytes of one plane of one bin
    const int ar = tid >> 3, ac = (tid & 7) * 4;          // A: 32 rows per pass, 8 lanes x 16 B = 16 complex per row
    const int h = lane >> 5, r32 = lane & 31;
    const int a_off = (wm * 32 + r32) * LDA + h * 16;
    const int b_off = DGRAD ? (h * 8) * 64 + wn * 32 + r32 : (wn * 32 + r32) * 20 + h * 8;
    // per-lane byte offsets (constants of the thread)
    const unsigned voA = (unsigned)(ar * Kc * 2 + ac) * 4u;                                  // + 32 rows for the second pass
    const unsigned voW = DGRAD ? (unsigned)((tid >> 4) * Nc + (tid & 15) * 4) * 4u : (unsigned)((tid >> 2) * Kc + (tid & 3) * 4) * 4u;
    const unsigned voC = (unsigned)((wm * 32 + 4 * h) * Nc + wn * 32 + r32) * 8u;
    const unsigned a_pass = (unsigned)(32 * Kc * 2) * 4u;
    const unsigned c_row = (unsigned)Nc * 8u;
    f32x4 ra[2], rb[3];
    // unit u = ((bin / 8 * MT + mt) * NT + nt) * 8 + (bin % 8): XCD j owns the bins = j (mod 8) and walks them bin-major with the
    // N-tiles of one M-tile back to back.  The stride G is a multiple of 8: its digits in (nt, mt, bin / 8) advance a unit.
    const int G8 = (int)gridDim.x >> 3;
    const int d_nt = G8 % NT, d_mt = (G8 / NT) % MT, d_b8 = G8 / (NT * MT);
    int u = blockIdx.x;
    if (u >= nunits) return;
    const int xcd = u & 7;
    int nt, mt, b8;
    { const int sq = u >> 3; nt = sq % NT; mt = (sq / NT) % MT; b8 = sq / (NT * MT); }
    __amdgpu_buffer_rsrc_t rsA, rsW, rsC, rsA_n, rsW_n;
    auto make_rs = [&](int bin, int m0, __amdgpu_buffer_rsrc_t& a_, __amdgpu_buffer_rsrc_t& w_) {
        a_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A) + ((size_t)bin * M + m0) * Kc * 2, 0, (int)((unsigned)(M - m0) * Kc * 8u), 0x00020000);
        w_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wf) + (size_t)bin * 3 * Nc * Kc, 0, (int)(3 * plane_b), 0x00020000);
    };
    auto gload = [&](const __amdgpu_buffer_rsrc_t& a_, const __amdgpu_buffer_rsrc_t& w_, int n0, int k0) {
        const unsigned soA = (unsigned)k0 * 8u;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
            ra[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_, voA + ps * a_pass, soA, 0));
        const unsigned soW = DGRAD ? (unsigned)(k0 * Nc + n0) * 4u : (unsigned)(n0 * Kc + k0) * 4u;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            rb[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_, voW, soW + p * plane_b, 0));
    };
    auto lstore = [&]() {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) *reinterpret_cast<f32x4*>(&As[(ps * 32 + ar) * LDA + ac]) = ra[ps];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (DGRAD) *reinterpret_cast<f32x4*>(&Bs[p * 16 * 64 + (tid >> 4) * 64 + (tid & 15) * 4]) = rb[p];
            else *reinterpret_cast<f32x4*>(&Bs[p * 64 * 20 + (tid >> 2) * 20 + (tid & 3) * 4]) = rb[p];
        }
    };
    make_rs(b8 * 8 + xcd, mt * 64, rsA, rsW);
    gload(rsA, rsW, nt * 64, 0);
    lstore();
    __syncthreads();
    for (;;) {
        const int bin = b8 * 8 + xcd, m0 = mt * 64, n0 = nt * 64;
        const int un = u + (int)gridDim.x;
        const bool has_next = un < nunits;
        int nt_n = nt + d_nt, mt_n = mt + d_mt, b8_n = b8 + d_b8;
        if (nt_n >= NT) { nt_n -= NT; ++mt_n; }
        if (mt_n >= MT) { mt_n -= MT; ++b8_n; }
        if (has_next) make_rs(b8_n * 8 + xcd, mt_n * 64, rsA_n, rsW_n);
        rsC = __builtin_amdgcn_make_buffer_rsrc(Cm + ((size_t)bin * M + m0) * Nc * 2, 0, (int)((unsigned)(M - m0) * Nc * 8u), 0x00020000);
        f32x16 acc1, acc2, acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
        for (int k0 = 0; k0 < Kc; k0 += 16) {
            const bool last = k0 + 16 >= Kc;
            if (!last) gload(rsA, rsW, n0, k0 + 16);
            else if (has_next) gload(rsA_n, rsW_n, nt_n * 64, 0);
            f32x4 b4[3][2];
            if (!DGRAD) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 2; ++q) b4[p][q] = *reinterpret_cast<const f32x4*>(&Bs[p * 64 * 20 + b_off + q * 4]);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&As[a_off + g * 4]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float xr = a4[2 * e], xi = a4[2 * e + 1];
                    float b1, b2, b3;
                    if (DGRAD) {
                        const int row = (2 * g + e) * 64;
                        b1 = Bs[b_off + row]; b2 = Bs[16 * 64 + b_off + row]; b3 = Bs[2 * 16 * 64 + b_off + row];
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xr + xi, b1, acc1, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, b2, acc2, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(xr, b3, acc3, 0, 0, 0);
                    } else {
                        const int q = g >> 1, idx = 2 * (g & 1) + e;
                        b1 = b4[0][q][idx]; b2 = b4[1][q][idx]; b3 = b4[2][q][idx];
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xr + xi, b1, acc1, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xr, b2, acc2, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, b3, acc3, 0, 0, 0);
                    }
                }
            }
            __syncthreads();
            if (!last || has_next) { lstore(); __syncthreads(); }
        }
        // accumulator register r = row (r & 3) + 8 (r >> 2) (+ 4h, in voC) of the wave's 32-row tile (in the per-lane offset: the
        // range check must see the row)
        const unsigned soC = (unsigned)n0 * 8u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x2 v = DGRAD ? f32x2{acc1[r] + acc2[r], acc1[r] - acc3[r]} : f32x2{acc1[r] - acc3[r], acc1[r] + acc2[r]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2_t, v), rsC, voC + (unsigned)((r & 3) + 8 * (r >> 2)) * c_row, soC, 0);
        }
        if (!has_next) break;
        u = un; nt = nt_n; mt = mt_n; b8 = b8_n; rsA = rsA_n; rsW = rsW_n;
    }
}

// icols: thread = (tile, kx, channel n): inverse FFT32 along ky, rows u < nrows kept.
// grid: x = tile group, y = kx; block = (256 / C) WHOLE tiles x all C channels (C = 64 << cq_shift <= 256): a block's loads are one
// contiguous 2 KB run per ky and its stores C x 8 B runs (round 6; before: 4 tiles x one 64-channel chunk, i.e. 512-byte pieces
// of 1-2 KB rows on the 128- / 256-channel layers -- PMC profiles/r06_fft_pmc.json: 1.8 TB/s, 70 % of the wave time waiting)
#ifndef GDN_ICOLS40_WAVES
#define GDN_ICOLS40_WAVES 2
#endif

template <int NP>
__global__ __launch_bounds__(256, NP == 40 ? GDN_ICOLS40_WAVES : 1) void ifft_cols_kernel(const float2* __restrict__ Yf, float2* __restrict__ S, int C, int M, int nrows,
                                                        int cq_shift) {
    constexpr int NK = FFT_NK_OF(NP);
    const int csh = 6 + cq_shift;                                   // log2(C)
    const int bx = GDN_ICOLS_ORDER ? blockIdx.y : blockIdx.x, kx = GDN_ICOLS_ORDER ? blockIdx.x : blockIdx.y;
    const int t = bx * (256 >> csh) + (threadIdx.x >> csh);
    if (t >= M) return;
    const int c = threadIdx.x & (C - 1);
    float re[NP], im[NP];
    // the block's (256 / C) tiles x C channels are one contiguous 2 KB run per bin: per-lane offset = 8 x thread, the bin row is scalar
    const int t0 = bx * (256 >> csh);
    const float2* src = fft_uni_ptr(Yf + ((size_t)kx * M + t0) * C);
    const size_t sk = (size_t)NK * M * C;
    const unsigned vi = threadIdx.x * 8u;
#pragma unroll
    for (int ky = 0; ky < NP; ++ky) {
        const f32x2 v = fft_ld2(src, vi);
        re[ky] = v[0]; im[ky] = v[1];
        src += sk;
    }
    fftn<NP, 1>(re, im);
    float2* dst = fft_uni_ptr(S + (size_t)t0 * fft_s_pitch(NP, C) +  (size_t)kx * C);
    const unsigned vo = (unsigned)(((size_t)(threadIdx.x >> csh) * fft_s_pitch(NP, C) + c) * 8);
    const int su = NK * C;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        if (u < nrows) fft_st2(dst, vo, re[u], im[u]);
        dst += su;
    }
}

// Reduction-over-tiles complex GEMM of the weight gradient:  dWf[bin][n][c] = sum_m conj(D[bin][m][n]) * X[bin][m][c]
// (D = spectrum of dy [M][N], X = spectrum of x [M][C]; both operands are read as they lie, rows = tiles), again with three
// real products:  g1 = dr (xr + xi), g2 = (dr + di) xr, g3 = (dr - di) xi;  re = g1 - g3, im = g1 - g2.
// 64 x 64 complex output tile, 16 tiles of the reduction per step; every MFMA operand pair is one conflict-free ds_read_b64.
// nsplit > 1: the reduction over the M tiles is cut into nsplit chunks (one workgroup each, partial spectra [split][bin][n][c]
// summed in fixed order by the tap kernel): with 64 channels there are only 544 (bin, tile) workgroups otherwise -- two per CU.
__global__ __launch_bounds__(256, 4) void cgemm_tn_bins_kernel(const float* __restrict__ D, const float* __restrict__ X,
                                                            float* __restrict__ dWf, int M, int N, int C, int nsplit,
                                                            int nbins) {
    constexpr int LD = 128;                              // 64 complex per row
    __shared__ __attribute__((aligned(16))) float Ds[16 * LD], Xs[16 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave >> 1, wj = wave & 1;
    const int TI = N / 64, TJ = C / 64;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int bin = (sq / (TI * TJ * nsplit)) * 8 + xcd, split = (sq / (TI * TJ)) % nsplit;
    const int i0 = ((sq / TJ) % TI) * 64, j0 = (sq % TJ) * 64;
    const int mchunk = ((M + nsplit - 1) / nsplit + 15) / 16 * 16;
    const int mb = split * mchunk, me = mb + mchunk < M ? mb + mchunk : M;        // this workgroup reduces tiles [mb, me)
    // raw buffer loads through descriptors of this workgroup's rows [mb, me) of the two spectra (columns from i0 / j0 on): the
    // per-lane offset is (row of the pass) x pitch + column, advanced by 16 rows per k-step; rows >= me fall outside the
    // descriptor and load as zero -- no predicate, no 64-bit address arithmetic (cgemm_bins_kernel above)
    const int nrow = me > mb ? me - mb : 0;
    __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(D) + (((size_t)bin * M + mb) * N + i0) * 2, 0,
                                                                   (int)((unsigned)nrow * N * 8u - (nrow ? (unsigned)i0 * 8u : 0u)), 0x00020000);
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) + (((size_t)bin * M + mb) * C + j0) * 2, 0,
                                                                   (int)((unsigned)nrow * C * 8u - (nrow ? (unsigned)j0 * 8u : 0u)), 0x00020000);
    f32x16 acc1, acc2, acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    const int lr = tid >> 5, lc = (tid & 31) * 4;         // 8 rows per pass, 32 lanes x 16 B per 512-byte row
    unsigned voD = (unsigned)(lr * N * 2 + lc) * 4u, voX = (unsigned)(lr * C * 2 + lc) * 4u;
    const unsigned passD = (unsigned)N * 64u, passX = (unsigned)C * 64u;          // 8 rows
    f32x4 rd[2], rx[2];
    auto gload = [&](int m0) {
        (void)m0;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            rd[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsD, voD + ps * passD, 0, 0));
            rx[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, voX + ps * passX, 0, 0));
        }
        voD += 2 * passD; voX += 2 * passX;
    };
    auto lstore = [&]() {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            *reinterpret_cast<f32x4*>(&Ds[(ps * 8 + lr) * LD + lc]) = rd[ps];
            *reinterpret_cast<f32x4*>(&Xs[(ps * 8 + lr) * LD + lc]) = rx[ps];
        }
    };
    const int d_off = (lane >> 5) * LD + (wi * 32 + (lane & 31)) * 2;
    const int x_off = (lane >> 5) * LD + (wj * 32 + (lane & 31)) * 2;
    gload(mb);
    lstore();
    __syncthreads();
    for (int m0 = mb; m0 < me; m0 += 16) {
        if (m0 + 16 < me) gload(m0 + 16);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float2 d = *reinterpret_cast<const float2*>(&Ds[d_off + kk * LD]);
            const float2 x = *reinterpret_cast<const float2*>(&Xs[x_off + kk * LD]);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d.x, x.x + x.y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(d.x + d.y, x.x, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(d.x - d.y, x.y, acc3, 0, 0, 0);
        }
        __syncthreads();
        if (m0 + 16 < me) { lstore(); __syncthreads(); }
    }
    float2* Pb = reinterpret_cast<float2*>(dWf) + ((size_t)split * nbins + bin) * N * C;
    const int col = j0 + wj * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Pb[(size_t)i * C + col] = make_float2(acc1[r] - acc3[r], acc1[r] - acc2[r]);
    }
}

// Inverse DFT of the weight-gradient spectrum dWf[bin][n][c] (complex) at the k*k taps only, separably:
//   g[ty] = sum_ky F[ky][kx] e^{+i 2 pi ky ty / NP}            (K complex values per kx)
//   dw[ty][tx] += alpha_kx Re( g[ty] e^{+i 2 pi kx tx / NP} )   (Hermitian weights 1 for kx = 0, NP/2, else 2)
// so the spectrum is read ONCE and a thread does NP*K + K*K complex MACs per kx instead of NP*K*K.
// Round 3: one WAVE per kx (was: a quarter of the kx bins per wave, N*C/64 workgroups in all -- 64 workgroups for a 64-channel
// layer, 388 us for 1.3 MB of output).  grid = (N*C/64, ceil(NK/4)): a workgroup is 64 (n, c) pairs x 4 kx, the ky loop is
// fully unrolled so every twiddle of the first stage is a literal (no scalar table lookups), the partial spectra of a split
// reduction GEMM are summed while they are loaded (fixed order; fft_sum_splits is gone), the four waves meet in LDS and the
// workgroup writes its partial tap set part[z * groups + y][tap][n][c]; fft_taps_reduce sums the partial sets in order.
// 40-point tiles (the 64-channel 9x9 layers: 384 workgroups otherwise) run with grid.z = 2: each workgroup transforms one half
// of the ky rows with literal twiddles of the LOCAL row index, the half's phase is one complex rotation of the K row sums
// (172 -> 127 us; the same split made the 32-point 7x7 kernel slower, 106 -> 157 us, and is not used there).
template <int NP>
__device__ __forceinline__ constexpr float tw_lit_cos(int q) {         // cos(2 pi q / NP), q in [0, NP): folded after unrolling
    constexpr float C40[11] = {1.0f, 0.98768834059513777f, 0.95105651629515353f, 0.89100652418836790f, 0.80901699437494745f,
                               0.70710678118654757f, 0.58778525229247314f, 0.45399049973954680f, 0.30901699437494745f,
                               0.15643446504023092f, 0.0f};
    constexpr float C32[9] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f,
                              0.55557023301960229f, 0.38268343236508984f, 0.19509032201612833f, 0.0f};
    if (NP == 40) return q <= 10 ? C40[q] : q <= 20 ? -C40[20 - q] : q <= 30 ? -C40[q - 20] : C40[40 - q];
    const int r = q * (32 / (NP == 40 ? 32 : NP));                   // angle in 32nds of a turn
    return r <= 8 ? C32[r] : r <= 16 ? -C32[16 - r] : r <= 24 ? -C32[r - 16] : C32[32 - r];
}
template <int NP>
__device__ __forceinline__ constexpr float tw_lit_sin(int q) { return tw_lit_cos<NP>((q + 3 * NP / 4) % NP); }   // sin x = cos(x - pi/2)

constexpr int FFT_TN_MAX_SPLITS = 4;    // reduction splits of the weight-gradient GEMM: the cap of tn_splits() AND the size of the tap kernel's load group

template <int K, int NP>
__global__ __launch_bounds__(256) void fft_wgrad_taps_kernel(const float* __restrict__ P, float* __restrict__ part, int N, int C,
                                                             int nsplit) {
    constexpr int NK = FFT_NK_OF(NP), BINS = FFT_BINS_OF(NP), KYH = NP == 40 ? NP / 2 : NP;   // 40-point tiles: blockIdx.z = which half of the ky rows
    __shared__ float red[3][K][64];                 // one filter row at a time: 7 KB, so the kernel never crowds a 107 KB
                                                    // transform workgroup of the other stream off a CU (the old 62 KB did)
    const int pl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + pl;             // (n, c) pair; N*C is a multiple of 64
    const int kx = blockIdx.y * 4 + grp;            // wave-uniform
    const size_t bs = (size_t)N * C;
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.f;
    if (kx < NK) {
        const int ky0 = blockIdx.z * KYH;
        const size_t sky = (size_t)NK * bs, ssp = (size_t)BINS * bs;
        const float2* F = reinterpret_cast<const float2*>(P) + (size_t)kx * bs + i + (size_t)ky0 * sky;
        float gr[K], gi[K];
#pragma unroll
        for (int ty = 0; ty < K; ++ty) { gr[ty] = 0.f; gi[ty] = 0.f; }
#pragma unroll
        for (int ky = 0; ky < KYH; ++ky) {            // (local row index: the half's phase e^{+i 2 pi ky0 ty / NP} is applied once, below)
            // P[0] + P[1] + ... in order, as fft_sum_splits did; at most four partial sets (tn_splits), loaded unconditionally
            // from clamped addresses so that the NP x 4 loads are all in flight instead of one dependent group per ky
            float2 u[FFT_TN_MAX_SPLITS];
#pragma unroll
            for (int sp = 0; sp < FFT_TN_MAX_SPLITS; ++sp) u[sp] = F[(size_t)(sp < nsplit ? sp : nsplit - 1) * ssp];
            float2 v = u[0];
#pragma unroll
            for (int sp = 1; sp < FFT_TN_MAX_SPLITS; ++sp)
                if (sp < nsplit) { v.x += u[sp].x; v.y += u[sp].y; }
            F += sky; GDN_KEEP(F);
#pragma unroll
            for (int ty = 0; ty < K; ++ty) {
                const float cs = tw_lit_cos<NP>((ky * ty) % NP), sn = tw_lit_sin<NP>((ky * ty) % NP);
                gr[ty] += v.x * cs - v.y * sn;
                gi[ty] += v.x * sn + v.y * cs;
            }
        }
        if (ky0) {
#pragma unroll
            for (int ty = 0; ty < K; ++ty) {
                const float cs = tw_cos<NP>(ky0 * ty), sn = tw_sin<NP>(ky0 * ty);
                const float a = gr[ty], b = gi[ty];
                gr[ty] = a * cs - b * sn;
                gi[ty] = a * sn + b * cs;
            }
        }
        const float alpha = (kx == 0 || kx == NP / 2) ? 1.f : 2.f;
#pragma unroll
        for (int tx = 0; tx < K; ++tx) {
            const float cs = tw_cos<NP>(kx * tx) * alpha, sn = tw_sin<NP>(kx * tx) * alpha;
#pragma unroll
            for (int ty = 0; ty < K; ++ty) acc[ty * K + tx] = gr[ty] * cs - gi[ty] * sn;
        }
    }
    float* dst = part + (size_t)(blockIdx.z * gridDim.y + blockIdx.y) * K * K * bs + i;
#pragma unroll
    for (int ty = 0; ty < K; ++ty) {
        if (ty > 0) __syncthreads();
        if (grp > 0) {
#pragma unroll
            for (int tx = 0; tx < K; ++tx) red[grp - 1][tx][pl] = acc[ty * K + tx];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int tx = 0; tx < K; ++tx)
                dst[(size_t)(ty * K + tx) * bs] = ((acc[ty * K + tx] + red[0][tx][pl]) + red[1][tx][pl]) + red[2][tx][pl];
        }
    }
}

// dw[tap][n][c] = scale * (part[0] + part[1] + ... + part[G-1]) (fixed order); n4 = K*K*N*C / 4
__global__ __launch_bounds__(256) void fft_taps_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int64_t n4, int G,
                                                              float scale) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(part)[i];
        for (int g = 1; g < G; ++g) v += reinterpret_cast<const f32x4*>(part)[(int64_t)g * n4 + i];
        reinterpret_cast<f32x4*>(dw)[i] = v * scale;
    }
}

// Data gradient, overlap-add: tile (ty, tx) of dy contributes a full 32x32 patch of dx at offset -pad.  The (at most two)
// tile rows that reach an image row are summed in the frequency domain before the row transform; along x the patches of
// even and odd tiles are written by two launches (parity): the first writer of a column stores (+ addsrc), the second adds.
template <int NP>
__global__ __launch_bounds__(256) void ifft_rows_overlap_kernel(const float2* __restrict__ S, float* __restrict__ dx, int lddx,
                                                                const float* __restrict__ addsrc, int ld_add, FftGeom g,
                                                                int parity, int Ho, int Wo, int off, int cq_shift) {
    // output image Ho x Wo; patch row j of tile row ty lands on output row ty*T - off + j  (off = pad for a zero-padded
    // layer: dx itself; off = 0 for a reflection-padded one: the padded-domain gradient, folded afterwards)
    // grid: x = group of tiles of this parity, y = output row, z = image; block = (256 / C) tiles x all C channels (whole 8 x C
    // byte rows of S in, whole 4 x C byte pixels of dx out: round 6, see ifft_cols_kernel)
    constexpr int NK = FFT_NK_OF(NP);
    const int C = g.C, T = g.T;
    const int ntx = (g.tiles_x + 1 - parity) / 2;         // tiles of this parity per row
    const bool remap = GDN_ROWS_REMAP == 1 || (GDN_ROWS_REMAP == 2 && cq_shift == 2);
    const int csh = remap ? 6 + cq_shift : 6;             // log2(channels per workgroup)
    const int txl = (remap ? blockIdx.x : (blockIdx.x >> cq_shift)) * (256 >> csh) + (threadIdx.x >> csh);
    if (txl >= ntx) return;
    const int c = (remap ? 0 : (blockIdx.x & ((1 << cq_shift) - 1)) * 64) + (threadIdx.x & ((1 << csh) - 1));
    const int tx = txl * 2 + parity, iy = blockIdx.y, b = blockIdx.z;
    const int q = iy + off;
    const int ty_a = q / T, j_a = q - ty_a * T;
    float re[NP], im[NP];
#pragma unroll
    for (int kx = 0; kx < NK; ++kx) { re[kx] = 0.f; im[kx] = 0.f; }
    if (ty_a < g.tiles_y) {
        const int t = (b * g.tiles_y + ty_a) * g.tiles_x + tx;
        const float2* src = S + (size_t)t * fft_s_pitch(NP, C) + ((size_t)j_a * NK) * C + c;
        const int skx = C;
#pragma unroll
        for (int kx = 0; kx < NK; ++kx) { const float2 v = *src; re[kx] = v.x; im[kx] = v.y; src += skx; GDN_KEEP(src); }
    }
    if (ty_a >= 1 && j_a + T < NP) {
        const int t = (b * g.tiles_y + ty_a - 1) * g.tiles_x + tx;
        const float2* src = S + (size_t)t * fft_s_pitch(NP, C) + ((size_t)(j_a + T) * NK) * C + c;
        const int skx = C;
#pragma unroll
        for (int kx = 0; kx < NK; ++kx) { const float2 v = *src; re[kx] += v.x; im[kx] += v.y; src += skx; GDN_KEEP(src); }
    }
#pragma unroll
    for (int kx = NK; kx < NP; ++kx) { re[kx] = re[NP - kx]; im[kx] = -im[NP - kx]; }
    fftn<NP, 1>(re, im);
    const int ix0 = tx * T - off;
    float* dst = dx + ((size_t)(b * Ho + iy) * Wo + ix0) * lddx + c;               // may point before the row: only
    const float* ad = addsrc ? addsrc + ((size_t)(b * Ho + iy) * Wo + ix0) * ld_add + c : nullptr;   // dereferenced in range
    const bool has_next = tx + 1 < g.tiles_x;
    const int km1 = g.k - 1;
    // (the pointers are made to depend on the transform's output: otherwise the loads below are hoisted above the transform,
    // their NP results stay live across it and the kernel drops from three waves per SIMD to two)
    asm volatile("" : "+v"(dst), "+v"(ad) : "v"(re[NP - 1]), "v"(re[0]));
    // Every column adds ONE earlier value to its result: what the even neighbours stored (odd tiles, shared columns), or the
    // addsrc term (first writer of a column), or nothing.  All of those loads are issued first, from clamped addresses, so
    // none sits behind a branch (with the read-modify-write inside the store loop the compiler waited for each before the
    // next: one dependent round trip per column).
    constexpr int HB = NP / 2;                 // (in two halves: NP / 2 loads in flight, NP / 2 extra registers)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        float prev[HB];
#pragma unroll
        for (int j2 = 0; j2 < HB; ++j2) {
            const int j = hb * HB + j2, ix = ix0 + j;
            const bool in = ix >= 0 && ix < Wo;
            const bool second = parity == 1 && (j < km1 || (j >= T && has_next));
            const int jc = (in ? j : (ix < 0 ? -ix0 : Wo - 1 - ix0)) - hb * HB;     // a column of this row that exists
            const float* src = (second || !ad) ? dst + (ptrdiff_t)jc * lddx : ad + (ptrdiff_t)jc * ld_add;
            prev[j2] = (second || ad) ? *src : 0.f;
        }
#pragma unroll
        for (int j2 = 0; j2 < HB; ++j2) {
            const int j = hb * HB + j2, ix = ix0 + j;
            if (ix >= 0 && ix < Wo) *dst = re[j] * (1.0f / (NP * NP)) + prev[j2];
            dst += lddx; GDN_KEEP(dst);
        }
        if (ad) { ad += (ptrdiff_t)HB * ld_add; GDN_KEEP(ad); }
    }
}

// dx[y][x] = sum of the padded-domain gradient over the padded coordinates that reflect onto (y, x)  (+ addsrc)
__global__ __launch_bounds__(256) void fft_reflect_fold_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int ldx,
                                                               const float* __restrict__ addsrc, int ld_add,
                                                               int B, int H, int W, int C, int p) {
    const int Hp = H + 2 * p, Wp = W + 2 * p, c4n = C / 4;
    const int64_t total = (int64_t)B * H * W * c4n;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H), b = (int)(t / H);
        int qy[3], qx[3], ny = 0, nx = 0;
        qy[ny++] = y + p;
        if (y >= 1 && y <= p) qy[ny++] = p - y;
        if (y <= H - 2 && y >= H - 1 - p) qy[ny++] = 2 * (H - 1) - y + p;
        qx[nx++] = x + p;
        if (x >= 1 && x <= p) qx[nx++] = p - x;
        if (x <= W - 2 && x >= W - 1 - p) qx[nx++] = 2 * (W - 1) - x + p;
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a)
            for (int e = 0; e < nx; ++e)
                s4 += *reinterpret_cast<const f32x4*>(dxp + ((size_t)(b * Hp + qy[a]) * Wp + qx[e]) * C + c4 * 4);
        const size_t op = (size_t)(b * H + y) * W + x;
        if (addsrc) s4 += *reinterpret_cast<const f32x4*>(addsrc + op * ld_add + c4 * 4);
        *reinterpret_cast<f32x4*>(dx + op * ldx + c4 * 4) = s4;
    }
}

// ---- single-pass 2-D transforms: one workgroup = one tile x 16 channels, rows and columns meet in LDS (68 KB) ----
// channels per workgroup: 16 (8 for the 40-point tiles -- 53.8 KB of LDS, two workgroups per CU at 152 registers -- measured again in
// round 6 with the scalar loader: forward transform 265 vs 237 us, inverse 357 vs 253: the 32- / 64-byte runs cost more than the
// overlap of two workgroups returns; profiles/r06_fft_ab.txt)
#define FFT_CG_OF(NP) 16
#define FFT_LDS_ELEMS_OF(NP) ((NP) * FFT_NK_OF(NP) * FFT_CG_OF(NP))

// One patch row of the forward transform's plain loader: NP raw buffer loads through `img` (image b, channel group: uniform), the
// lane's row as a constant byte offset (FFT_VOFF_ZERO for a row that reads as zero), the column as the scalar offset.
//   FAST: no border column in this tile -- the column is ix0 + bb, columns >= nvalid (zero padding of a dy tile) read 0;
//   else: bounds test, reflection (|ix|, 2W - 2 - ix) and the zero-byte descriptor of an outside column, all scalar.
//   AFF : [relu](x * scale + shift) on load (the producer's train-mode BatchNorm); padding stays zero -- isl / itl are already
//         zeroed for a zero row, the shift of a zero column by a scalar select.
template <int NP, bool FAST, bool AFF>
__device__ __forceinline__ void fft_load_row(float (&re)[NP], float (&im)[NP], const float* img, unsigned img_bytes, unsigned vrow,
                                             int ldx, int ix0, int nvalid, int lim, int W, float isl, float itl, float lo) {
    const unsigned ldx4 = (unsigned)ldx * 4u;
#pragma unroll
    for (int bb = 0; bb < NP; ++bb) {
        const int ix = ix0 + bb;
        bool colok;
        unsigned so;
        if (FAST) {
            colok = bb < nvalid;
            so = (unsigned)ix * ldx4;
        } else {
            colok = bb < nvalid && ix >= -lim && ix < W + lim;
            const int ai = ix < 0 ? -ix : ix;
            so = (unsigned)(ai < W ? ai : 2 * W - 2 - ai) * ldx4;                       // mirrored column (border patches only)
        }
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fft_rs(img, colok ? img_bytes : 0u), vrow, so, 0));
        re[bb] = AFF ? fmaxf(v * isl + (colok ? itl : 0.f), lo) : v;
        im[bb] = 0.f;
    }
}

// forward: patch (halo = 1: rows/cols start at -pad, full 32; halo = 0: the T x T tile, zero padded) -> Xf[bin][tile][C]
// two waves per SIMD: at four (128 VGPRs) the two 32-point transforms spill 44 dwords per lane and the kernel is 15 % slower
// in_scale != NULL: the tensor read is the RAW output of the producer convolution and the layer input is
// [relu](x * in_scale[c] + in_shift[c]) -- the producer's train-mode BatchNorm (+ReLU) applied on load, so that
// activation is never written to memory (ResidualBlock AE_model_unet.py:49-54).  Padding stays zero.
// bnb_y != NULL (halo = 0, the dy transform of a backward): x is dout, the gradient of z = [relu](BN_train(bnb_y)), and
// the tile transformed is this layer's dy = scale*(dz - k1 - xhat*k2), dz = dout*[z>0] (pass 3 of the BatchNorm backward,
// AE_model_unet.py:51,54) computed on load from bnb_co = {scale, shift, mean, invstd}[C] and bnb_kk = {k1, k2}[C]: dy is
// never written to memory.
template <int NP>
__global__ __launch_bounds__(NP * FFT_CG_OF(NP), NP == 16 ? 4 : 2) void fft2d_fwd_kernel(const float* __restrict__ x, int ldx, float2* __restrict__ Xf,
                                                        FftGeom g, int halo, const float* __restrict__ in_scale,
                                                        const float* __restrict__ in_shift, int in_relu,
                                                        const float* __restrict__ bnb_y, int ld_bnb,
                                                        const float* __restrict__ bnb_co, const float* __restrict__ bnb_kk,
                                                        int up2x) {
    constexpr int NK = FFT_NK_OF(NP), CG = FFT_CG_OF(NP), CGS = CG == 8 ? 3 : 4;
    __shared__ float2 lds[FFT_LDS_ELEMS_OF(NP)];
    // XCD-aware order: XCD j (= blockIdx & 7) owns the contiguous tile range [j, j+1) * ceil(M/8) and runs the channel
    // groups of one tile back to back, so the half cache lines the groups share and the halo rows / columns neighbouring
    // tiles share are served by that XCD's L2 instead of being fetched once per XCD
    const int ngrp = g.C / CG, tpx = (g.M + 7) / 8;
    const int tid = threadIdx.x, c = tid & (CG - 1), cg = GDN_UNI(((blockIdx.x >> 3) % ngrp) * CG);
    const int t = GDN_UNI((blockIdx.x & 7) * tpx + (blockIdx.x >> 3) / ngrp);
    if (t >= g.M) return;
    const int tx = GDN_UNI(t % g.tiles_x), ty = GDN_UNI((t / g.tiles_x) % g.tiles_y), b = GDN_UNI(t / (g.tiles_x * g.tiles_y));
    float re[NP], im[NP];
    {
        const int a = tid >> CGS;
        const int iy = ty * g.T + a - (halo ? g.pad : 0);
        const int ix0 = tx * g.T - (halo ? g.pad : 0);
        const int nvalid = halo ? NP : g.T;
        // reflection border (halo patches only): rows / columns -pad..-1 and H..H+pad-1 mirror the image; anything
        // further out only feeds outputs beyond the image and reads as zero
        const int lim = (halo && g.reflect) ? g.pad : 0;
        const bool row_ok = iy >= -lim && iy < g.H + lim && a < nvalid;
        const int iyr = iy < 0 ? -iy : (iy >= g.H ? 2 * g.H - 2 - iy : iy);
        // wave-uniform image base + 32-bit per-lane offsets (one image is far below 2^31 elements): half the address registers
        const float* img = fft_uni_ptr(x + (size_t)b * g.H * g.W * ldx + cg);
        const int row_off = (row_ok ? iyr : 0) * g.W * ldx + c;
        int off_x = ix0 * ldx;                             // running ix * ldx (interior columns: no multiply per element)
        const float is = in_scale ? in_scale[cg + c] : 1.f, it = in_scale ? in_shift[cg + c] : 0.f;
        const float lo = in_relu ? 0.f : -3.402823466e38f;
        if (a >= nvalid) {
            // zero-padding rows of a dy tile (whole waves when T is a multiple of four rows): nothing to load or transform
#pragma unroll
            for (int kx = 0; kx < NK; ++kx) { re[kx] = 0.f; im[kx] = 0.f; }
        } else {
        if (up2x) {
            // x is the LOW-resolution tensor [B][H/2][W/2][ldx]; the layer input is its x2 bilinear upsampling
            // (up2x = 1: align_corners False, 2: True), interpolated here from the four neighbours of every element --
            // border rule first, on the upsampled coordinates (up2x.h)
            const int Hl = g.H >> 1, Wl = g.W >> 1, align = up2x - 1;
            float ly; int y0, y1;
            up_src(row_ok ? iyr : 0, Hl, align, ly, y0, y1);
            const float* lo_img = x + (size_t)b * Hl * Wl * ldx + cg + c;
            const float* r0 = lo_img + (size_t)y0 * Wl * ldx;
            const float* r1 = lo_img + (size_t)y1 * Wl * ldx;
#pragma unroll
            for (int bb = 0; bb < NP; ++bb) {
                const int ix = ix0 + bb;
                const bool ok = row_ok && bb < nvalid && ix >= -lim && ix < g.W + lim;
                const int ixr = ix < 0 ? -ix : (ix >= g.W ? 2 * g.W - 2 - ix : ix);
                float v = 0.f;
                if (ok) {
                    float lx; int x0, x1;
                    up_src(ixr, Wl, align, lx, x0, x1);
                    v = up2x_at(r0, r1, ldx, ly, x0, x1, lx);
                }
                re[bb] = v;
                im[bb] = 0.f;
            }
        } else if (!bnb_y) {
            // Round 6: raw buffer loads, no branch and no predicate per element.  The ROW is a constant of the thread: a row that must
            // read as zero gets a per-lane offset no descriptor reaches (the hardware returns 0).  The COLUMN is uniform over the
            // workgroup: its bounds test, reflection and byte offset are scalar arithmetic on the descriptor's base; a column outside
            // the border reads a clamped address and is zeroed by a scalar select.  (Before: a divergent branch, 22 scalar and 7
            // vector instructions around every load -- 1200 of the 1900 instructions a wave spent in this phase.)
            // A column outside the border is read through a descriptor of zero bytes (it loads 0 without touching memory).  A tile
            // with no border column (11 of the 13 tiles of a 416-pixel row) takes the loop whose column is a running scalar offset.
            const unsigned vrow = row_ok ? (unsigned)row_off * 4u : FFT_VOFF_ZERO;
            const unsigned img_bytes = (unsigned)g.H * g.W * ldx * 4u - (unsigned)cg * 4u;
            const float isl = row_ok ? is : 0.f, itl = row_ok ? it : 0.f;      // (the coefficients of a zero ROW are zeroed per lane)
            const bool fastx = ix0 >= 0 && ix0 + nvalid <= g.W;                // uniform
            if (fastx) {
                if (in_scale) fft_load_row<NP, true, true>(re, im, img, img_bytes, vrow, ldx, ix0, nvalid, lim, g.W, isl, itl, lo);
                else fft_load_row<NP, true, false>(re, im, img, img_bytes, vrow, ldx, ix0, nvalid, lim, g.W, isl, itl, lo);
            } else {
                if (in_scale) fft_load_row<NP, false, true>(re, im, img, img_bytes, vrow, ldx, ix0, nvalid, lim, g.W, isl, itl, lo);
                else fft_load_row<NP, false, false>(re, im, img, img_bytes, vrow, ldx, ix0, nvalid, lim, g.W, isl, itl, lo);
            }
        } else {
            // dy = scale * (dz - k1 - xhat * k2) from (dout, y): zero border (no reflection in this mode).
            // Round 6: raw buffer loads like the plain loader -- the lane's row is a constant offset (a zero row reads 0 and its
            // scale is zeroed per lane), the column is uniform (scalar offset; an outside column reads through a zero-byte
            // descriptor and is re-selected to 0 by a scalar mask).  Two halves: NP loads in flight, NP / 2 extra registers.
            const int ch = cg + c;
            const float bs = bnb_co[ch], bt = bnb_co[g.C + ch], bmu = bnb_co[2 * g.C + ch], bis = bnb_co[3 * g.C + ch];
            const float k1 = bnb_kk[ch], k2 = bnb_kk[g.C + ch];
            const float bsl = row_ok ? bs : 0.f;
            const float* yimg = fft_uni_ptr(bnb_y + (size_t)b * g.H * g.W * ld_bnb + cg);
            const unsigned vrow_d = row_ok ? (unsigned)row_off * 4u : FFT_VOFF_ZERO;
            const unsigned vrow_y = row_ok ? (unsigned)(iy * g.W * ld_bnb + c) * 4u : FFT_VOFF_ZERO;
            const unsigned bytes_d = (unsigned)g.H * g.W * ldx * 4u - (unsigned)cg * 4u;
            const unsigned bytes_y = (unsigned)g.H * g.W * ld_bnb * 4u - (unsigned)cg * 4u;
            const unsigned ldx4 = (unsigned)ldx * 4u, ldy4 = (unsigned)ld_bnb * 4u;
            constexpr int HB = NP / 2;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                float dv[HB];
#pragma unroll
                for (int b2 = 0; b2 < HB; ++b2) {
                    const int bb = hb * HB + b2, ix = ix0 + bb;
                    const bool colok = bb < nvalid && ix >= 0 && ix < g.W;              // uniform
                    dv[b2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fft_rs(img, colok ? bytes_d : 0u), vrow_d,
                                                                                            (unsigned)ix * ldx4, 0));
                    im[bb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fft_rs(yimg, colok ? bytes_y : 0u), vrow_y,
                                                                                            (unsigned)ix * ldy4, 0));   // (im[] holds y for now)
                }
#pragma unroll
                for (int b2 = 0; b2 < HB; ++b2) {
                    const int bb = hb * HB + b2, ix = ix0 + bb;
                    const bool colok = bb < nvalid && ix >= 0 && ix < g.W;
                    const float d = dv[b2], yv = im[bb];
                    const float dz = (in_relu && !(yv * bs + bt > 0.f)) ? 0.f : d;
                    const float val = bsl * (dz - k1 - ((yv - bmu) * bis) * k2);
                    re[bb] = colok ? val : 0.f;
                    im[bb] = 0.f;
                }
            }
        }
        fftn<NP, -1>(re, im);
        }
#pragma unroll
        for (int kx = 0; kx < NK; ++kx) lds[(a * NK + kx) * CG + c] = make_float2(re[kx], im[kx]);
    }
    __syncthreads();
    if (tid < NK * CG) {
        const int kx = tid >> CGS;
#pragma unroll
        for (int a = 0; a < NP; ++a) { const float2 v = lds[(a * NK + kx) * CG + c]; re[a] = v.x; im[a] = v.y; }
        fftn<NP, -1>(re, im);
        // bin (ky, kx) of this tile and channel group: the tile / channel-group / ky part of the address is uniform (descriptor
        // base), the lane's kx and channel are a constant 32-bit offset
        float2* dst = fft_uni_ptr(Xf + (size_t)t * g.C + cg);
        const size_t sk = (size_t)NK * g.M * g.C;
        const unsigned vo = (unsigned)(((size_t)kx * g.M * g.C + c) * 8);
#pragma unroll
        for (int ky = 0; ky < NP; ++ky) { fft_st2(dst, vo, re[ky], im[ky]); dst += sk; }
    }
}

// inverse, first half shared by both consumers: Yf[bin][tile][C] -> lds[u][kx][c] (inverse along ky)
template <int NP>
__device__ __forceinline__ void ifft2d_cols_to_lds(const float2* __restrict__ Yf, float2* lds, int C, int M, int t, int cg,
                                                   float (&re)[NP], float (&im)[NP]) {
    constexpr int NK = FFT_NK_OF(NP), CG = FFT_CG_OF(NP), CGS = CG == 8 ? 3 : 4;
    const int tid = threadIdx.x, c = tid & (CG - 1);
    if (tid < NK * CG) {
        const int kx = tid >> CGS;
        const float2* src = Yf + ((size_t)kx * M + t) * C + cg + c;
        const size_t sk = (size_t)NK * M * C;
#pragma unroll
        for (int ky = 0; ky < NP; ++ky) {
            const float2 v = *src;
            re[ky] = v.x; im[ky] = v.y;
            src += sk; GDN_KEEP(src);
        }
        fftn<NP, 1>(re, im);
#pragma unroll
        for (int u = 0; u < NP; ++u) lds[(u * NK + kx) * CG + c] = make_float2(re[u], im[u]);
    }
    __syncthreads();
}

template <int NP>
__device__ __forceinline__ void ifft_row_from_lds(const float2* lds, int u, int c, float (&re)[NP], float (&im)[NP]) {
    constexpr int NK = FFT_NK_OF(NP), CG = FFT_CG_OF(NP), CGS = CG == 8 ? 3 : 4;
#pragma unroll
    for (int kx = 0; kx < NK; ++kx) { const float2 v = lds[(u * NK + kx) * CG + c]; re[kx] = v.x; im[kx] = v.y; }
#pragma unroll
    for (int kx = NK; kx < NP; ++kx) { re[kx] = re[NP - kx]; im[kx] = -im[NP - kx]; }
    fftn<NP, 1>(re, im);
}

// forward convolution output: valid T x T outputs of the tile, epilogue, BatchNorm partials (stats slot = tile)
template <int NP>
__global__ __launch_bounds__(NP * FFT_CG_OF(NP), NP == 16 ? 8 : NP == 32 ? 4 : 2) void ifft2d_valid_kernel(const float2* __restrict__ Yf, float* __restrict__ y, int ldy,
                                                           const float* __restrict__ addsrc, int ld_add,
                                                           float* __restrict__ stats, const float* __restrict__ ep_scale,
                                                           const float* __restrict__ ep_shift, int act, FftGeom g) {
    constexpr int CG = FFT_CG_OF(NP), CGS = CG == 8 ? 3 : 4;
    __shared__ float2 lds[FFT_LDS_ELEMS_OF(NP)];
    const int ngrp = g.N / CG, tpx = (g.M + 7) / 8;          // XCD-aware order as in fft2d_fwd_kernel
    const int tid = threadIdx.x, c = tid & (CG - 1), cg = GDN_UNI(((blockIdx.x >> 3) % ngrp) * CG), T = g.T;
    const int t = GDN_UNI((blockIdx.x & 7) * tpx + (blockIdx.x >> 3) / ngrp);
    if (t >= g.M) return;
    const int tx = GDN_UNI(t % g.tiles_x), ty = GDN_UNI((t / g.tiles_x) % g.tiles_y), b = GDN_UNI(t / (g.tiles_x * g.tiles_y));
    float re[NP], im[NP];
    ifft2d_cols_to_lds<NP>(Yf, lds, g.N, g.M, t, cg, re, im);
    const int u = tid >> CGS, oy = ty * T + u;
    float s1 = 0.f, s2 = 0.f;
    if (u < T && oy < g.H) {
        ifft_row_from_lds<NP>(lds, u, c, re, im);
        const int ox0 = tx * T;
        // (pointer walks here: the buffer-descriptor form of these loads / stores measured 4 % slower, profiles/r06_fft_ab.txt)
        float* dst = y + ((size_t)(b * g.H + oy) * g.W + ox0) * ldy + cg + c;
        const float* ad = addsrc ? addsrc + ((size_t)(b * g.H + oy) * g.W + ox0) * ld_add + cg + c : nullptr;
        if (!ep_scale && act == 0 && !ad) {
            // training forward (raw conv output + BatchNorm partials): nothing but the scale, the sums and the store
#pragma unroll
            for (int v = 0; v < NP; ++v) {
                if (v < T && ox0 + v < g.W) {
                    const float val = re[v] * (1.0f / (NP * NP));
                    s1 += val; s2 += val * val;
                    *dst = val;
                }
                dst += ldy; GDN_KEEP(dst);
            }
        } else if (ep_scale && !(act & GDN_ACT_TANH)) {
            // eval-mode layer (frozen guide, inference): folded BatchNorm, optional ReLU, optional residual
            const float es = ep_scale[cg + c], et = ep_shift[cg + c];
            const float lo = (act & GDN_ACT_RELU) ? 0.f : -3.402823466e38f;
#pragma unroll
            for (int v = 0; v < NP; ++v) {
                if (v < T && ox0 + v < g.W) {
                    const float raw = re[v] * (1.0f / (NP * NP));
                    s1 += raw; s2 += raw * raw;
                    float val = fmaxf(raw * es + et, lo);
                    if (ad) val += *ad;
                    *dst = val;
                }
                dst += ldy; GDN_KEEP(dst);
                if (ad) { ad += ld_add; GDN_KEEP(ad); }
            }
        } else {
            const float es = ep_scale ? ep_scale[cg + c] : 1.f, et = ep_shift ? ep_shift[cg + c] : 0.f;
#pragma unroll
            for (int v = 0; v < NP; ++v) {
                if (v < T && ox0 + v < g.W) {
                    float val = re[v] * (1.0f / (NP * NP));
                    s1 += val; s2 += val * val;
                    if (ep_scale) val = val * es + et;
                    if (act & GDN_ACT_RELU) val = fmaxf(val, 0.f);
                    if (ad) val += *ad;
                    if (act & GDN_ACT_TANH) val = tanhf(val);
                    *dst = val;
                }
                dst += ldy; GDN_KEEP(dst);
                if (ad) { ad += ld_add; GDN_KEEP(ad); }
            }
        }
    }
    if (stats) {
        __syncthreads();                       // everyone is done reading the spectrum rows
        float* red = reinterpret_cast<float*>(lds);
        red[tid * 2] = s1; red[tid * 2 + 1] = s2;
        __syncthreads();
        if (tid < CG) {
            float a1 = 0.f, a2 = 0.f;
            for (int j = 0; j < NP; ++j) { a1 += red[(j * CG + tid) * 2]; a2 += red[(j * CG + tid) * 2 + 1]; }
            stats[((size_t)t * 2 + 0) * g.N + cg + tid] = a1;
            stats[((size_t)t * 2 + 1) * g.N + cg + tid] = a2;
        }
    }
}

// ---- data gradient, round 6: single-pass inverse into a PATCH buffer + gather ------------------------------------------------
// The two-kernel inverse (ifft_cols -> S -> ifft_rows_overlap x 2) moves the 447 MB intermediate S out and back in, and its column
// pass runs at 1.8-2.6 TB/s (profiles/r06_fft_pmc.json).  Inside a training step the backward is HBM-throughput-bound (running the
// weight-gradient chain beside the data-gradient chain returns 0.3 ms of its 6.8: the memory is saturated either way), so bytes are
// what count.  Here the tile's whole NP x NP patch of dx contributions is produced in one pass (inverse columns -> LDS -> Hermitian
// inverse rows, like ifft2d_valid) and written to patch[tile][u][v][C] -- no two tiles write the same address, so ONE launch and no
// read-modify-write -- and a streaming gather sums the <= 4 patches that cover an output pixel in a fixed order (+ addsrc).
// Bytes per 9x9 layer at B = 20: 447 (E) + 426 (patch) written once and read once + 272 (addsrc) + 272 (dx) = 1.84 GB against ~2.5.
template <int NP>
__global__ __launch_bounds__(NP * FFT_CG_OF(NP), NP == 16 ? 8 : NP == 32 ? 4 : 2) void ifft2d_patch_kernel(const float2* __restrict__ Ef, float* __restrict__ patch,
                                                                                                              FftGeom g, int Ho, int Wo, int off) {
    constexpr int CG = FFT_CG_OF(NP), CGS = CG == 8 ? 3 : 4;
    __shared__ float2 lds[FFT_LDS_ELEMS_OF(NP)];
    const int ngrp = g.C / CG, tpx = (g.M + 7) / 8;          // XCD-aware order as in fft2d_fwd_kernel
    const int tid = threadIdx.x, c = tid & (CG - 1), cg = GDN_UNI(((blockIdx.x >> 3) % ngrp) * CG);
    const int t = GDN_UNI((blockIdx.x & 7) * tpx + (blockIdx.x >> 3) / ngrp);
    if (t >= g.M) return;
    const int tx = GDN_UNI(t % g.tiles_x), ty = GDN_UNI((t / g.tiles_x) % g.tiles_y);
    float re[NP], im[NP];
    ifft2d_cols_to_lds<NP>(Ef, lds, g.C, g.M, t, cg, re, im);
    const int u = tid >> CGS, iy = ty * g.T - off + u;
    if (iy < 0 || iy >= Ho) return;                          // (after the barrier inside ifft2d_cols_to_lds: no later one)
    ifft_row_from_lds<NP>(lds, u, c, re, im);
    const int ix0 = tx * g.T - off;
    float* dst = patch + (size_t)t * fft_patch_tile(NP, g.C) + (size_t)u * fft_patch_row(NP, g.C) + cg + c;
#pragma unroll
    for (int v = 0; v < NP; ++v) {
        if (ix0 + v >= 0 && ix0 + v < Wo) *dst = re[v] * (1.0f / (NP * NP));      // (uniform bound: a scalar branch)
        dst += g.C; GDN_KEEP(dst);
    }
}

// dx[b][y][x][c] = (patch of the tile whose rows / columns start at or before the pixel) + (the previous tile's, where its k - 1
// trailing rows / columns reach the pixel), rows first, in that fixed order, + addsrc.  A workgroup walks image rows r = b Ho + y
// (r = blockIdx.x, + gridDim.x, ...); a thread owns four channels (fixed) of every (256 / (C / 4))-th pixel of the row; the tile /
// patch-column of every x is tabulated once per workgroup in LDS (no division per pixel).
// bnb_y != NULL: dx is the final gradient of z = [relu](BN_train(bnb_y)) (this layer's input): the workgroup also emits that
// BatchNorm's backward partial sums (sum dz, sum dz * xhat; bnb_co = {scale, shift, mean, invstd}[C]) into bnb_part[blockIdx.x][2][C]
// -- the values are in registers here, so the stand-alone reduce pass over (dx, y) (bn_bwd_reduce: two tensor reads) becomes one
// read of y.  Fixed order everywhere: a thread's pixels in sequence, the threads of a channel group in index order.
constexpr int FFT_GATHER_MAX_W = 1024;       // widest output row the x table holds (wider rows: gdn_fftconv_bwd takes the two-kernel inverse)
constexpr int FFT_GATHER_MAX_BLOCKS = 8192;  // workgroups (= BatchNorm-backward partial slots) of one gather launch
__global__ __launch_bounds__(256) void fft_overlap_gather_kernel(const float* __restrict__ patch, float* __restrict__ dx, int lddx,
                                                                 const float* __restrict__ addsrc, int ld_add, FftGeom g, int np,
                                                                 int Ho, int Wo, int off, int c4_shift,
                                                                 const float* __restrict__ bnb_y, int ld_bnb,
                                                                 const float* __restrict__ bnb_co, int bnb_relu,
                                                                 float* __restrict__ bnb_part) {
    __shared__ int xinfo[FFT_GATHER_MAX_W];
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x, c4n = 1 << c4_shift, c4 = tid & (c4n - 1), xl = tid >> c4_shift, xstep = 256 >> c4_shift;
    const int T = g.T, C = g.C;
    for (int x = tid; x < Wo; x += 256) {
        const int qx = x + off, txa = qx / T;
        xinfo[x] = (txa << 8) | (qx - txa * T);              // (patch column < T <= 36, tile column < 2^23)
    }
    __syncthreads();
    f32x4 bsc = {0.f, 0.f, 0.f, 0.f}, bsh = bsc, bmu = bsc, bis = bsc;
    if (bnb_y) {
        bsc = *reinterpret_cast<const f32x4*>(bnb_co + c4 * 4);
        bsh = *reinterpret_cast<const f32x4*>(bnb_co + C + c4 * 4);
        bmu = *reinterpret_cast<const f32x4*>(bnb_co + 2 * C + c4 * 4);
        bis = *reinterpret_cast<const f32x4*>(bnb_co + 3 * C + c4 * 4);
    }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    const size_t ttile = fft_patch_tile(np, C), trow = fft_patch_row(np, C);      // floats per tile / per patch row (padded)
    const int nrows = g.B * Ho;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        const int b = r / Ho, y = r - b * Ho;
        const int qy = y + off, tya = qy / T, ja = qy - tya * T;
        const bool ya = tya < g.tiles_y, yb = tya >= 1 && ja + T < np;           // candidate tile rows (uniform)
        const float* base = patch + (size_t)b * g.tiles_y * g.tiles_x * ttile + (size_t)c4 * 4;
        const float* rowa = base + (size_t)tya * g.tiles_x * ttile + (size_t)ja * trow;
        const float* rowb = base + (size_t)(tya - 1) * g.tiles_x * ttile + (size_t)(ja + T) * trow;
        for (int x = xl; x < Wo; x += xstep) {
            const int info = xinfo[x], txa = info >> 8, ia = info & 255;
            const bool xa = txa < g.tiles_x, xb = txa >= 1 && ia + T < np;
            const size_t oa = (size_t)txa * ttile + (size_t)ia * C, ob = (size_t)(txa - 1) * ttile + (size_t)(ia + T) * C;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (ya) {
                if (xa) s += *reinterpret_cast<const f32x4*>(rowa + oa);
                if (xb) s += *reinterpret_cast<const f32x4*>(rowa + ob);
            }
            if (yb) {
                if (xa) s += *reinterpret_cast<const f32x4*>(rowb + oa);
                if (xb) s += *reinterpret_cast<const f32x4*>(rowb + ob);
            }
            const size_t px = (size_t)r * Wo + x;
            if (addsrc) s += *reinterpret_cast<const f32x4*>(addsrc + px * ld_add + c4 * 4);
            *reinterpret_cast<f32x4*>(dx + px * lddx + c4 * 4) = s;
            if (bnb_y) {
                const f32x4 yv = *reinterpret_cast<const f32x4*>(bnb_y + px * ld_bnb + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dz = (bnb_relu && !(yv[e] * bsc[e] + bsh[e] > 0.f)) ? 0.f : s[e];
                    s1[e] += dz;
                    s2[e] += dz * ((yv[e] - bmu[e]) * bis[e]);
                }
            }
        }
    }
    if (bnb_part) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = s1[e]; red[tid * 8 + 4 + e] = s2[e]; }
        __syncthreads();
        if (tid < c4n) {
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
            for (int k = 0; k < xstep; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { a1[e] += red[(k * c4n + tid) * 8 + e]; a2[e] += red[(k * c4n + tid) * 8 + 4 + e]; }
            float* dst = bnb_part + (size_t)blockIdx.x * 2 * C + tid * 4;
            *reinterpret_cast<f32x4*>(dst) = a1;
            *reinterpret_cast<f32x4*>(dst + C) = a2;
        }
    }
}

bool fft_geom(const gdn_conv_geom* g, FftGeom& f) {
    if (!g || g->stride != 1 || g->k < 3 || g->k > 9 || (g->k & 1) == 0) return false;
    if (g->pad != g->k / 2 || (g->transposed && g->pad_mode != 0)) return false;
    if (g->pad_mode == 1 && (g->pad >= g->H || g->pad >= g->W)) return false;
    if ((g->Cin % 64) || (g->Cout % 64) || g->Cin > 256 || g->Cout > 256) return false;
    f.B = g->B; f.H = g->H; f.W = g->W; f.C = g->Cin; f.N = g->Cout; f.k = g->k; f.pad = g->pad;
    // 16-point tiles for a small window when the WEIGHT spectrum would outweigh the activations' at 32 points (544 bins x 3
    // planes x Cin x Cout floats against tiles x 544 x Cin complex: 1.5 Cout > tiles), i.e. wide layers on small images --
    // the 5x5 / 256-channel blocks at 32 x 104: the weight spectrum is 4x smaller, the level tiles with T = 12 without the
    // waste of T = 28, the per-bin GEMMs get M = 540 rows instead of 160 (measured: forward GEMM -31 %, step -1.7 ms).
    {
        const int t32 = 33 - g->k;
        const long m32 = (long)gdn_plan_batch(g) * cdiv(g->H, t32) * cdiv(g->W, t32);
        f.np = (g->k <= 5 && 3L * g->Cout > 2 * m32) ? 16 : 32;
        // 40-point tiles for a TRAINED layer (GDN_HINT_TRAIN) where they cut the transformed points by at least a fifth -- the
        // 9x9 layers at 128 x 416: T = 32 tiles the image exactly, 4 x 13 x 1600 points against 6 x 18 x 1024 -- and the weight
        // spectrum stays small (<= 128 channels).  Measured at B = 20 (tests/diag/fft_kernels_time.py): the three per-bin GEMMs
        // of a training step -26 %, the single-pass transforms +20-35 % (one 640-thread workgroup per CU): forward 0.85 ->
        // 0.81 ms, backward 1.33 -> 1.17 ms; an eval-mode forward (frozen guide, inference) is slower and keeps 32 points.
        const int t40 = 41 - g->k;
        const long p32 = (long)cdiv(g->H, t32) * cdiv(g->W, t32) * 1024, p40 = (long)cdiv(g->H, t40) * cdiv(g->W, t40) * 1600;
        const int force = (g->hints & GDN_HINT_FFT_NP32) ? 32 : (g->hints & GDN_HINT_FFT_NP40) ? 40 : 0;   // measurement / test override
        if (f.np == 32 && (g->hints & GDN_HINT_TRAIN) && g->k >= 7 && g->Cin <= 128 && g->Cout <= 128 && 5 * p40 <= 4 * p32) f.np = 40;
        if (force == 32 && f.np == 40) f.np = 32;
        if (force == 40 && f.np == 32 && g->k >= 5) f.np = 40;
    }
    f.bins = FFT_BINS_OF(f.np);
    f.T = f.np - g->k + 1;
    f.tiles_y = cdiv(g->H, f.T); f.tiles_x = cdiv(g->W, f.T);
    f.M = g->B * f.tiles_y * f.tiles_x;
    f.reflect = g->pad_mode == 1;
    f.flip = g->transposed ? 1 : 0;
    return true;
}

inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }
inline dim3 icols_grid(const FftGeom& f, int cq_shift) {
    const int groups = cdiv(f.M, 4 >> cq_shift), nk = f.np / 2 + 1;
    return GDN_ICOLS_ORDER ? dim3(nk, groups) : dim3(groups, nk);
}
// persistent grid of cgemm_bins_kernel: four workgroups per CU, ROUNDED DOWN to a multiple of 8 (at least 8) -- the kernel's
// mixed-radix walk takes its stride digits from gridDim.x >> 3 (a workgroup stays on one XCD's bins), so a grid that is not a
// multiple of 8 (a part with an odd CU count) would skip or repeat units
inline int cgemm_grid(int nunits) {
    int g = 4 * gdn_num_cus() / 8 * 8;
    if (g < 8) g = 8;
    return nunits < g ? (nunits + 7) / 8 * 8 : g;
}

// splits of the weight-gradient reduction: enough workgroups for ~4 per CU, chunks of at least 64 tiles
inline int tn_splits(const FftGeom& f) {
    const int wgs = (f.N / 64) * (f.C / 64) * f.bins;
    int s = (4096 + wgs - 1) / wgs;
    if (s > FFT_TN_MAX_SPLITS) s = FFT_TN_MAX_SPLITS;
    while (s > 1 && f.M / s < 64) --s;
    return s < 1 ? 1 : s;
}
// weight-spectrum / weight-gradient-product region of the backward workspace: the weight planes (data gradient without a saved
// state), or the per-split products of the reduction GEMM followed by the partial tap sets of fft_wgrad_taps
inline size_t tn_prod_bytes(const FftGeom& f) { return al256((size_t)tn_splits(f) * f.bins * 2 * f.C * f.N * 4); }
inline int taps_kyh(const FftGeom& f) { return f.np == 40 ? 2 : 1; }                // ky halves (fft_wgrad_taps: 40-point tiles only)
inline int taps_groups(const FftGeom& f) { return taps_kyh(f) * cdiv(f.np / 2 + 1, 4); }     // partial tap sets: (kx group) x (ky half)
inline size_t wf_region_bytes(const FftGeom& f) {
    const size_t planes = (size_t)f.bins * 3 * f.C * f.N * 4;
    const size_t prod = tn_prod_bytes(f) + al256((size_t)taps_groups(f) * f.k * f.k * f.C * f.N * 4);
    return al256(planes > prod ? planes : prod);
}

}  // namespace

// workspace: Xf, Yf (M*544*C / N complex), Wf (544 * 3 * N * C floats)
extern "C" size_t gdn_fftconv_fwd_workspace_bytes(const gdn_conv_geom* g) {
    FftGeom f;
    if (!fft_geom(g, f)) return 0;
    return al256((size_t)f.M * f.bins * f.C * 8) + al256((size_t)f.M * f.bins * f.N * 8) +
           al256((size_t)f.bins * 3 * f.C * f.N * 4);
}

extern "C" int64_t gdn_fftconv_stats_slots(const gdn_conv_geom* g) {
    FftGeom f;
    if (!fft_geom(g, f)) return GDN_ERR_UNSUPPORTED;
    return f.M;          // one slot per tile
}

// saved state of one forward for its backward: input spectrum Xf, then the weight spectrum Wf
extern "C" size_t gdn_fftconv_spectrum_bytes(const gdn_conv_geom* g) {
    FftGeom f;
    if (!fft_geom(g, f)) return 0;
    return al256((size_t)f.M * f.bins * f.C * 8) + al256((size_t)f.bins * 3 * f.C * f.N * 4);
}

namespace {
template <int NP>
void launch_weights_np(const FftGeom& f, const float* w, float* Wf, hipStream_t st) {
    const dim3 gr(cdiv(f.N * f.C, 256), NP);
    switch (f.k) {
        case 3: hipLaunchKernelGGL((fft_weights_kernel<3, NP>), gr, dim3(256), 0, st, w, Wf, f.N, f.C, f.flip); break;
        case 5: hipLaunchKernelGGL((fft_weights_kernel<5, NP>), gr, dim3(256), 0, st, w, Wf, f.N, f.C, f.flip); break;
        case 7: hipLaunchKernelGGL((fft_weights_kernel<7, NP>), gr, dim3(256), 0, st, w, Wf, f.N, f.C, f.flip); break;
        default: hipLaunchKernelGGL((fft_weights_kernel<9, NP>), gr, dim3(256), 0, st, w, Wf, f.N, f.C, f.flip); break;
    }
}
void launch_weights(const FftGeom& f, const float* w, float* Wf, hipStream_t st) {
    if (f.np == 16) launch_weights_np<16>(f, w, Wf, st);
    else if (f.np == 40) launch_weights_np<40>(f, w, Wf, st);
    else launch_weights_np<32>(f, w, Wf, st);
}

// the transform kernels, dispatched on the tile size
void launch_fft2d_fwd(const FftGeom& f, int chans, const float* x, int ldx, float2* Xf, int halo, const float* in_scale,
                      const float* in_shift, int in_relu, const float* bnb_y, int ld_bnb, const float* bnb_co, const float* bnb_kk,
                      int up2x, hipStream_t st) {
    const dim3 gr(chans / FFT_CG_OF(f.np) * 8 * cdiv(f.M, 8)), bl(f.np * FFT_CG_OF(f.np));
    if (f.np == 16)
        hipLaunchKernelGGL(fft2d_fwd_kernel<16>, gr, bl, 0, st, x, ldx, Xf, f, halo, in_scale, in_shift, in_relu, bnb_y,
                           ld_bnb, bnb_co, bnb_kk, up2x);
    else if (f.np == 40)
        hipLaunchKernelGGL(fft2d_fwd_kernel<40>, gr, bl, 0, st, x, ldx, Xf, f, halo, in_scale, in_shift, in_relu, bnb_y,
                           ld_bnb, bnb_co, bnb_kk, up2x);
    else
        hipLaunchKernelGGL(fft2d_fwd_kernel<32>, gr, bl, 0, st, x, ldx, Xf, f, halo, in_scale, in_shift, in_relu, bnb_y,
                           ld_bnb, bnb_co, bnb_kk, up2x);
}
}  // namespace

extern "C" int gdn_fftconv_fwd(const gdn_conv_geom* g, const float* x, int32_t ldx, const float* w, float* y, int32_t ldy,
                               const float* addsrc, int32_t ld_add, float* stats, const float* ep_scale,
                               const float* ep_shift, int32_t act, const float* in_scale, const float* in_shift,
                               int32_t in_relu, int32_t in_up2x, void* xf_out, void* workspace, size_t workspace_bytes,
                               void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    FftGeom f;
    if (!fft_geom(g, f)) return GDN_ERR_UNSUPPORTED;
    if (!x || !w || !y || (!ep_scale) != (!ep_shift) || (!in_scale) != (!in_shift)) return GDN_ERR_BAD_ARG;
    if (in_up2x < 0 || in_up2x > 2 || (in_up2x && (in_scale || (g->H & 1) || (g->W & 1)))) return GDN_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gdn_fftconv_fwd_workspace_bytes(g)) return GDN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)workspace;
    float2* Xf = (float2*)p; p += al256((size_t)f.M * f.bins * f.C * 8);
    float2* Yf = (float2*)p; p += al256((size_t)f.M * f.bins * f.N * 8);
    float* Wf = (float*)p;
    if (xf_out) {
        Xf = (float2*)xf_out;
        Wf = (float*)((char*)xf_out + al256((size_t)f.M * f.bins * f.C * 8));
    }
    launch_fft2d_fwd(f, f.C, x, ldx, Xf, 1, in_scale, in_shift, in_relu, nullptr, 0, nullptr, nullptr, in_up2x, st);
    launch_weights(f, w, Wf, st);
    {
        const int nu = cdiv(f.M, 64) * (f.N / 64) * f.bins;
        hipLaunchKernelGGL(cgemm_bins_kernel<false>, dim3(cgemm_grid(nu)), dim3(256), 0, st,
                           (const float*)Xf, (const float*)Wf, (float*)Yf, f.M, f.N, f.C, nu);
    }
    const dim3 gv(f.N / FFT_CG_OF(f.np) * 8 * cdiv(f.M, 8)), bv(f.np * FFT_CG_OF(f.np));
    if (f.np == 16)
        hipLaunchKernelGGL(ifft2d_valid_kernel<16>, gv, bv, 0, st, (const float2*)Yf, y, ldy, addsrc, ld_add, stats,
                           ep_scale, ep_shift, act, f);
    else if (f.np == 40)
        hipLaunchKernelGGL(ifft2d_valid_kernel<40>, gv, bv, 0, st, (const float2*)Yf, y, ldy, addsrc, ld_add, stats,
                           ep_scale, ep_shift, act, f);
    else
        hipLaunchKernelGGL(ifft2d_valid_kernel<32>, gv, bv, 0, st, (const float2*)Yf, y, ldy, addsrc, ld_add, stats,
                           ep_scale, ep_shift, act, f);
    return gdn_launch_status();
}

// workspace: R/S intermediate, Df (spectrum of dy), Ef (spectrum of the dx patches), Wf / dWf (544 * 3 * N * C floats)
extern "C" size_t gdn_fftconv_bwd_workspace_bytes(const gdn_conv_geom* g) {
    FftGeom f;
    if (!fft_geom(g, f)) return 0;
    if (f.flip) return 0;                  // stride-1 ConvTranspose2d: forward (inference) only
    if (f.C != 64 && f.C != 128 && f.C != 256) return 0;   // the inverse kernels of the data gradient index channels by shifts: a
                                                           // caller asking "is the backward supported" (ops.fft_ok) takes another path
    const size_t cm = f.C > f.N ? f.C : f.N;
    const size_t padded = f.reflect ? al256((size_t)f.B * (f.H + 2 * f.pad) * (f.W + 2 * f.pad) * f.C * 4) : 0;
    return al256((size_t)f.M * fft_s_pitch(f.np, (int)cm) * 8) + al256((size_t)f.M * f.bins * f.N * 8) +
           al256((size_t)f.M * f.bins * f.C * 8) + wf_region_bytes(f) + padded;
}

// workgroups of the data gradient's gather pass = slots of the BatchNorm-backward partial sums it can emit (0: not available:
// reflection-padded layers finish their gradient in the fold pass, and rows wider than the gather's x table take the old inverse)
static int fft_gather_blocks(const FftGeom& f) {
    if (!GDN_FFT_DGRAD_PATCH || f.reflect || f.W > FFT_GATHER_MAX_W) return 0;
    const int64_t rows = (int64_t)f.B * f.H;
    return (int)(rows < FFT_GATHER_MAX_BLOCKS ? rows : FFT_GATHER_MAX_BLOCKS);
}
extern "C" int64_t gdn_fftconv_bnb_slots(const gdn_conv_geom* g) {
    FftGeom f;
    if (!fft_geom(g, f) || f.flip) return 0;
    if (f.C != 64 && f.C != 128 && f.C != 256) return 0;
    return fft_gather_blocks(f);
}

extern "C" int gdn_fftconv_bwd(const gdn_conv_geom* g, const float* dy, int32_t ldy, const float* w, const void* xf,
                               float* dx, int32_t ldx, const float* addsrc, int32_t ld_add, float* dw,
                               const float* dyb_y, int32_t ld_dyb, const float* dyb_co,
                               const float* dyb_kk, int32_t dyb_relu,
                               const float* bnb_y, int32_t ld_bnb, const float* bnb_co, int32_t bnb_relu, float* bnb_partial,
                               int32_t dx_up2x, int32_t phases, void* workspace,
                               size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    FftGeom f;
    if (!fft_geom(g, f) || f.flip) return GDN_ERR_UNSUPPORTED;
    if (dyb_y && (!dyb_co || !dyb_kk)) return GDN_ERR_BAD_ARG;
    if (bnb_y && (!dx || !bnb_co || !bnb_partial || (ld_bnb % 4))) return GDN_ERR_BAD_ARG;
    if (bnb_y && (fft_gather_blocks(f) == 0 || (ldx % 4) || (addsrc && (ld_add % 4)))) return GDN_ERR_UNSUPPORTED;
    if (!dy || (!dx && !dw) || (dx && !w && !xf) || (dw && !xf)) return GDN_ERR_BAD_ARG;
    if (dx && f.reflect && ((ldx % 4) || (addsrc && (ld_add % 4)))) return GDN_ERR_UNSUPPORTED;
    if (dx && f.C != 64 && f.C != 128 && f.C != 256) return GDN_ERR_UNSUPPORTED;      // (the inverse kernels index channels by shifts)
    if (dx_up2x < 0 || dx_up2x > 2 || (dx_up2x && ((g->H & 1) || (g->W & 1)))) return GDN_ERR_BAD_ARG;
    if (dx && dx_up2x && !f.reflect) return GDN_ERR_UNSUPPORTED;   // (the fold pass of a reflection layer carries the adjoint)
    if (!workspace || workspace_bytes < gdn_fftconv_bwd_workspace_bytes(g)) return GDN_ERR_WORKSPACE;
    if (phases == 0) phases = GDN_FFT_BWD_ALL;
    hipStream_t st = (hipStream_t)stream;
    const size_t cm = f.C > f.N ? f.C : f.N;
    char* p = (char*)workspace;
    float2* R = (float2*)p; p += al256((size_t)f.M * fft_s_pitch(f.np, (int)cm) * 8);
    float2* Df = (float2*)p; p += al256((size_t)f.M * f.bins * f.N * 8);
    float2* Ef = (float2*)p; p += al256((size_t)f.M * f.bins * f.C * 8);
    float* Wf = (float*)p; p += wf_region_bytes(f);   // weight-gradient products P (per split), or the weight spectrum when the forward saved none
    float* dxp = (float*)p;          // reflection layers: gradient over the padded domain
    auto blocks = [](int64_t n) { const int64_t b = cdiv64(n, 256); return (unsigned)(b < 65536 * 8 ? b : 65536 * 8); };
    if (phases & GDN_FFT_BWD_TRANSFORM) {
        // spectrum of the dy tiles (no halo: rows / columns >= T are the zero padding of the linear convolution)
        FftGeom fd = f;
        fd.C = f.N;
        fd.reflect = 0;
        launch_fft2d_fwd(fd, f.N, dy, ldy, Df, 0, nullptr, nullptr, dyb_relu, dyb_y, ld_dyb, dyb_co, dyb_kk, 0, st);
    }
    // The weight-gradient chain (reduction GEMM + tap transform) and the data-gradient chain only share Df and use disjoint
    // parts of the workspace, so a caller may run them as separate calls on two streams of its own (phases; the library
    // itself holds no stream, event or other state).
    if (dw && (phases & GDN_FFT_BWD_DW)) {
        float* P = Wf;
        const int ns = tn_splits(f);
        if (ns > FFT_TN_MAX_SPLITS) return GDN_ERR_UNSUPPORTED;          // (the tap kernel sums at most that many partial sets)
        hipLaunchKernelGGL(cgemm_tn_bins_kernel, dim3((f.N / 64) * (f.C / 64) * f.bins * ns), dim3(256), 0, st,
                           (const float*)Df, (const float*)xf, P, f.M, f.N, f.C, ns, f.bins);
        float* part = (float*)((char*)P + tn_prod_bytes(f));
        const dim3 gt(f.N * f.C / 64, taps_groups(f) / taps_kyh(f), taps_kyh(f));
#define GDN_TAPS(KK) case KK: \
            if (f.np == 16) hipLaunchKernelGGL((fft_wgrad_taps_kernel<KK, 16>), gt, dim3(256), 0, st, (const float*)P, part, f.N, f.C, ns); \
            else if (f.np == 40) hipLaunchKernelGGL((fft_wgrad_taps_kernel<KK, 40>), gt, dim3(256), 0, st, (const float*)P, part, f.N, f.C, ns); \
            else hipLaunchKernelGGL((fft_wgrad_taps_kernel<KK, 32>), gt, dim3(256), 0, st, (const float*)P, part, f.N, f.C, ns); \
            break;
        switch (f.k) {
            GDN_TAPS(3) GDN_TAPS(5) GDN_TAPS(7) GDN_TAPS(9)
        }
        const int64_t n4 = (int64_t)f.k * f.k * f.N * f.C / 4;
        hipLaunchKernelGGL(fft_taps_reduce_kernel, dim3(blocks(n4)), dim3(256), 0, st, (const float*)part, dw, n4, taps_groups(f),
                           1.0f / (f.np * f.np));
#undef GDN_TAPS
    }
    if (dx && (phases & GDN_FFT_BWD_DX)) {
        const float* Wsaved = xf ? (const float*)((const char*)xf + al256((size_t)f.M * f.bins * f.C * 8)) : nullptr;
        if (!Wsaved) launch_weights(f, w, Wf, st);
        const int nu = cdiv(f.M, 64) * (f.C / 64) * f.bins;
        hipLaunchKernelGGL(cgemm_bins_kernel<true>, dim3(cgemm_grid(nu)), dim3(256), 0, st,
                           (const float*)Df, Wsaved ? Wsaved : (const float*)Wf, (float*)Ef, f.M, f.C, f.N, nu);
        const int Ho = f.reflect ? f.H + 2 * f.pad : f.H, Wo = f.reflect ? f.W + 2 * f.pad : f.W;
        float* o = f.reflect ? dxp : dx;
        const int ldo = f.reflect ? f.C : ldx, off = f.reflect ? 0 : f.pad;
        const float* ad = f.reflect ? (const float*)nullptr : addsrc;
        if (GDN_FFT_DGRAD_PATCH && (ldo % 4) == 0 && (!ad || (ld_add % 4) == 0) && Wo <= FFT_GATHER_MAX_W) {
            // single-pass inverse of every tile into its own patch (the region S used to occupy), then the gather
            float* patch = (float*)R;
            const dim3 gp(f.C / FFT_CG_OF(f.np) * 8 * cdiv(f.M, 8)), bp(f.np * FFT_CG_OF(f.np));
            if (f.np == 16) hipLaunchKernelGGL(ifft2d_patch_kernel<16>, gp, bp, 0, st, (const float2*)Ef, patch, f, Ho, Wo, off);
            else if (f.np == 40) hipLaunchKernelGGL(ifft2d_patch_kernel<40>, gp, bp, 0, st, (const float2*)Ef, patch, f, Ho, Wo, off);
            else hipLaunchKernelGGL(ifft2d_patch_kernel<32>, gp, bp, 0, st, (const float2*)Ef, patch, f, Ho, Wo, off);
            int c4_shift = 4;                                    // log2(C / 4): C is 64, 128 or 256
            while ((4 << c4_shift) < f.C) ++c4_shift;
            const int64_t rows = (int64_t)f.B * Ho;
            const int gb = (int)(rows < FFT_GATHER_MAX_BLOCKS ? rows : FFT_GATHER_MAX_BLOCKS);      // (= fft_gather_blocks(f) when bnb_y)
            hipLaunchKernelGGL(fft_overlap_gather_kernel, dim3(gb), dim3(256), 0, st, (const float*)patch, o, ldo,
                               ad, ld_add, f, f.np, Ho, Wo, off, c4_shift, bnb_y, ld_bnb, bnb_co, bnb_relu, bnb_partial);
        } else {
        // inverse along ky into S, then rows: the tile rows that reach an image row are summed in the frequency domain
        int cq_shift = 0;
        while ((64 << cq_shift) < f.C) ++cq_shift;           // C / 64 is 1, 2 or 4
        if (f.np == 16)
            hipLaunchKernelGGL(ifft_cols_kernel<16>, icols_grid(f, cq_shift), dim3(256), 0, st, (const float2*)Ef, R,
                               f.C, f.M, 16, cq_shift);
        else if (f.np == 40)
            hipLaunchKernelGGL(ifft_cols_kernel<40>, icols_grid(f, cq_shift), dim3(256), 0, st, (const float2*)Ef, R,
                               f.C, f.M, 40, cq_shift);
        else
            hipLaunchKernelGGL(ifft_cols_kernel<32>, icols_grid(f, cq_shift), dim3(256), 0, st, (const float2*)Ef, R,
                               f.C, f.M, 32, cq_shift);
        for (int parity = 0; parity < 2; ++parity) {
            const int ntx = (f.tiles_x + 1 - parity) / 2;
            if (ntx == 0) continue;
            const bool remap = GDN_ROWS_REMAP == 1 || (GDN_ROWS_REMAP == 2 && cq_shift == 2);
            const dim3 gr(remap ? cdiv(ntx, 4 >> cq_shift) : cdiv(ntx, 4) << cq_shift, Ho, f.B);
            if (f.np == 16)
                hipLaunchKernelGGL(ifft_rows_overlap_kernel<16>, gr, dim3(256), 0, st, (const float2*)R, o, ldo, ad, ld_add, f, parity, Ho,
                                   Wo, off, cq_shift);
            else if (f.np == 40)
                hipLaunchKernelGGL(ifft_rows_overlap_kernel<40>, gr, dim3(256), 0, st, (const float2*)R, o, ldo, ad, ld_add, f, parity, Ho,
                                   Wo, off, cq_shift);
            else
                hipLaunchKernelGGL(ifft_rows_overlap_kernel<32>, gr, dim3(256), 0, st, (const float2*)R, o, ldo, ad, ld_add, f, parity, Ho,
                                   Wo, off, cq_shift);
        }
        }
        if (f.reflect && dx_up2x)
            // dx is the gradient of the LOW-resolution tensor the forward upsampled on load: fold + adjoint interpolation
            hipLaunchKernelGGL(reflect_fold_up2x_kernel, dim3(blocks((int64_t)f.B * (f.H / 2) * (f.W / 2) * (f.C / 4))), dim3(256), 0,
                               st, (const void*)dxp, (void*)dx, ldx, (const void*)addsrc, ld_add, f.B, f.H, f.W, f.C, f.pad, dx_up2x - 1, 0);
        else if (f.reflect)
            hipLaunchKernelGGL(fft_reflect_fold_kernel, dim3(blocks((int64_t)f.B * f.H * f.W * (f.C / 4))), dim3(256), 0, st,
                               (const float*)dxp, dx, ldx, addsrc, ld_add, f.B, f.H, f.W, f.C, f.pad);
    }
    return gdn_launch_status();
}

// Measurement hook (bench.py roofline_cgemm): ONLY the per-bin complex GEMMs of a frequency-domain layer, on
// whatever the workspace holds -- which: 0 the forward's cgemm_bins<false> (Y = X W), 1 the data gradient's cgemm_bins<true>
// (E = D conj-transposed W), 2 the weight gradient's reduction cgemm_tn_bins (P = sum over tiles D^H X).  Three real products per
// complex product (Gauss).  workspace: gdn_fftconv_cgemm_workspace_bytes(); 1 and 2 read the same D, so a caller may run them
// on two streams of its own as the backward does.
extern "C" size_t gdn_fftconv_cgemm_workspace_bytes(const gdn_conv_geom* g) {
    FftGeom f;
    if (!fft_geom(g, f)) return 0;
    const size_t cm = f.C > f.N ? f.C : f.N;
    return 3 * al256((size_t)f.M * f.bins * cm * 8) + wf_region_bytes(f) + al256((size_t)f.bins * 3 * f.C * f.N * 4);
}
extern "C" int gdn_fftconv_cgemm(const gdn_conv_geom* g, int32_t which, void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    FftGeom f;
    if (!fft_geom(g, f) || which < 0 || which > 2) return GDN_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < gdn_fftconv_cgemm_workspace_bytes(g)) return GDN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t cm = f.C > f.N ? f.C : f.N, sp = al256((size_t)f.M * f.bins * cm * 8);
    char* p = (char*)workspace;
    const float* A = (const float*)p;                   // X (0) / D (1, 2)
    float* O = (float*)(p + (which == 0 ? sp : 2 * sp));    // Y / E: 1 and 2 may run side by side (E and P are different regions)
    const float* X2 = (const float*)(p + sp);           // (2) the saved input spectrum
    float* P = (float*)(p + 3 * sp);
    const float* Wf = (const float*)(p + 3 * sp + wf_region_bytes(f));
    if (which == 0) {
        const int nu = cdiv(f.M, 64) * (f.N / 64) * f.bins;
        hipLaunchKernelGGL(cgemm_bins_kernel<false>, dim3(cgemm_grid(nu)), dim3(256), 0, st, A, Wf, O, f.M, f.N, f.C, nu);
    } else if (which == 1) {
        const int nu = cdiv(f.M, 64) * (f.C / 64) * f.bins;
        hipLaunchKernelGGL(cgemm_bins_kernel<true>, dim3(cgemm_grid(nu)), dim3(256), 0, st, A, Wf, O, f.M, f.C, f.N, nu);
    }
    else {
        const int ns = tn_splits(f);
        if (ns > FFT_TN_MAX_SPLITS) return GDN_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(cgemm_tn_bins_kernel, dim3((f.N / 64) * (f.C / 64) * f.bins * ns), dim3(256), 0, st, A, X2, P, f.M, f.N, f.C, ns, f.bins);
    }
    return gdn_launch_status();
}
// host query: {bins, tiles M, Gauss real products}: the GEMM shape behind the hook above
extern "C" int gdn_fftconv_cgemm_shape(const gdn_conv_geom* g, int32_t* bins, int32_t* M, int32_t* np) {
    FftGeom f;
    if (!fft_geom(g, f)) return GDN_ERR_UNSUPPORTED;
    if (bins) *bins = f.bins;
    if (M) *M = f.M;
    if (np) *np = f.np;
    return GDN_OK;
}
