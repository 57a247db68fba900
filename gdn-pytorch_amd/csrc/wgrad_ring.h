// wgrad_ring_bf16: the bf16 weight gradient of the stride-1 layers with a k x k window, k >= 3, on the LDS-DMA structure
// (round 5; included by conv_wgrad_bf16.hip).  Replaces autograd's conv backward-weights for ResidualBlock / ConvBlock,
// /root/reference/src/AE_model_unet.py:50,53,67 (loss.backward(), trainer.py:467,767).
//
//   dW[ky][kx][co][ci] = sum_{b,oy,ox} dY[b][oy][ox][co] * X[b][oy - p + ky][ox - p + kx][ci]
//
// The round-1 kernel (conv_wgrad_bf16 below in the .hip) stages both operands global -> registers -> LDS, puts two barriers
// around a 13-k-step segment, reads two transposing LDS fragments per MFMA and needs two resident workgroups per CU to hide
// all that: 0.29-0.37 of the bf16 pipe.  This kernel:
//
//   * ONE persistent 512-thread workgroup per CU owns a PAIR of filter rows (ky0, ky0 + 1) of one 64 x 64 (co, ci) channel
//     tile and a contiguous range of STAGES; a stage is `nr` whole output rows (or a <= 208-pixel piece of one row) of one
//     image.  Waves: 2 (filter row z) x 2 (co half) x 2 (ci half); a wave keeps the 32 x 32 tiles of all KW taps of its
//     filter row in registers (KW x 16 accumulators).  Both filter rows read the same dY image, and the X rows they need
//     overlap: filter row z of output row oy reads input row oy + ky0 + z - p, so with one-row stages X rows live in a RING
//     of three row slots and every stage brings in ONE new X row and one dY row for 2 x KW x 4 tile-taps of MFMA work
//     (9x9: 54 KB per 7488 MFMA cycles).
//   * Everything is staged by LDS-DMA (buffer_load_dwordx4 ... lds), one stage ahead, into [pixel][64 channels] images
//     with 128-byte rows -- whole lines on the global side -- whose 64-byte halves are swapped on pixels with bit 1 set:
//     the four pixel rows a 16-lane group of ds_read_b64_tr_b16 transposes then fall on four different 64-byte bank
//     windows (conflict-free at every alignment).  The swap is applied to the per-lane SOURCE address of the DMA and again
//     on the read.  ONE raw s_barrier per stage (thousands of MFMA cycles), nothing else synchronises.
//   * The KW taps of a filter row are shifted windows of the same X row.  A lane's k-slots are 8 CONSECUTIVE pixels, so
//     the fragment of tap kx is the 16-pixel window of the lane shifted by kx elements: even taps are register quads of the
//     window as it was read, odd taps the same after one v_alignbit per register -- 6 transposing reads and 7 VALU
//     instructions per 9 MFMAs instead of 20 reads.
//   * Split-K over stages: slab `split` of the workspace holds this workgroup's raw fp32 sums, wgrad_bf16_reduce_kernel
//     adds the slabs in slab order (bitwise reproducible: the stage -> split map is a function of the geometry only).
#pragma once

#define WR_GPX 224                        // pixels of a dY stage image (14 k-steps)
#define WR_GBYTES (WR_GPX * 128)
#define WR_XBYTES (88 * 1024)             // X row slots: ring of 3 rows, or 2 images of nr + 1 rows
#define WR_X0 (2 * WR_GBYTES)
#define WR_TAB (WR_X0 + WR_XBYTES)
#define WR_DUMMY (WR_TAB + 128)           // where the LDS-DMA pieces that do not exist land
#define WR_LDS (WR_DUMMY + 1024)

struct WgRingParams {
    const void* g; const void* x; float* part;
    int B, Hg, Wg, ldg, Cg;
    int Hx, Wx, ldx, Cx;
    int k, pad, pad_mode;
    int n_cgt, n_cxt;
    int S, ips, nst;                      // splits, stages per split, stages in all
    int TW, TWp, nstrip, nr, sph;         // strip width, padded to 8, strips per row, rows per stage, stages per strip column
    int RP, nks, ring;                    // X row pitch in positions, k-steps per stage, X rows live in a ring (nr == 1)
    unsigned g_bytes, x_bytes;
    int gmagic, xmagic;                   // 65536 / TWp + 1, 65536 / (RP / 8) + 1: the kernel's divisions by multiplication
    int knobs;                            // measurement knobs (0 in production): 1 no DMA after the first stage, 2 no k-loop, 4 bare stage boundary
};

typedef int wr_i32x4 __attribute__((ext_vector_type(4)));

#if defined(__HIP_DEVICE_COMPILE__)
#define WR_DEVICE_BODY 1
#else
#define WR_DEVICE_BODY 0
#endif

// registers of the two X windows a k-step reads (pixel pairs): taps 4m / 4m+1 take w[2m..2m+3] / the same shifted by one
// element (needs w[2m+4]); taps 4m+2 / 4m+3 take the window read two positions further on, w2[2m..2m+3] / shifted
template <int KW> struct WrWin {
    static constexpr int nw() { int n = 0; for (int kx = 0; kx < KW; ++kx) if ((kx & 3) < 2) { const int v = 2 * (kx >> 2) + 4 + (kx & 1); n = v > n ? v : n; } return n; }
    static constexpr int nw2() { int n = 0; for (int kx = 0; kx < KW; ++kx) if ((kx & 3) >= 2) { const int v = 2 * (kx >> 2) + 4 + (kx & 1); n = v > n ? v : n; } return n; }
};

typedef int wr_i32x8 __attribute__((ext_vector_type(8)));

// the scheduling pattern of a k-step: KW groups of {one MFMA, its share of the NRD LDS reads, vector ALU, scalar ALU}
// (the builtin wants literal group sizes)
template <int KW, int NRD, int I> struct WrSched {
    static __device__ __forceinline__ void apply() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        // the reads go out two per MFMA, behind the FIRST MFMAs of the step: the last of them still has (KW - NRD / 2) MFMAs of
        // this wave to land in before the next step's first MFMA wants it
        if constexpr (2 * I + 1 < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        else if constexpr (2 * I < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);
        if constexpr (I + 1 < KW) WrSched<KW, NRD, I + 1>::apply();
    }
};

template <int KW, bool RING>
__global__ __launch_bounds__(512, 2) void wgrad_ring_bf16(const WgRingParams p) {
#if WR_DEVICE_BODY
    constexpr int NW = WrWin<KW>::nw(), NW2 = WrWin<KW>::nw2();
    constexpr int NR1 = (NW + 1) / 2, NR2 = (NW2 + 1) / 2;     // transposing reads per window (4 positions = 2 registers each)
    __shared__ __attribute__((aligned(16))) unsigned char sm[WR_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int z = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
    const int h = lane >> 5, q = (lane >> 2) & 3, g16 = (lane >> 4) & 1, pp = lane & 3;

    // ---- this workgroup's unit.  Workgroups b, b + 8, ... share an XCD; ids are dealt so that the workgroups of one split (same
    // dY rows, overlapping X rows) are neighbours there.  The workgroup of the single last filter row of an odd window has half
    // of its waves idle (10 % of the 9x9 launch, 12.5 % of 7x7, 17 % of 5x5).  Measured and not kept, both with such a workgroup
    // taking TWO splits so that the others get fewer stages: (a) as it is -- LDS-DMA brings ~25 GB/s into a CU, a stage's 54 KB
    // take 2.2 us, and with one row's MFMA work per stage those workgroups became DMA-bound and the critical path: 9x9 0.70 ->
    // 0.91 ms; (b) its two wave groups sharing the k-steps of the one row (summed through LDS at the end): a stage still costs it
    // 3.3 us against 5.8 us of a row pair's (barrier, first reads, the DMA pieces its 6-7 k-steps cannot cover): 0.62 -> 0.75 ms.
    const int nblk = (int)gridDim.x;
    const int id = (int)(blockIdx.x & 7) * (nblk >> 3) + (int)(blockIdx.x >> 3);
    const int nct = p.n_cgt * p.n_cxt, nkg = (p.k + 1) >> 1;
    const int wps = nkg * nct;                                 // workgroups per split
    const int split0 = id / wps, r = id - split0 * wps;
    if (split0 >= p.S) return;
    const int kg = r % nkg, ct = r / nkg;
    const int nsplit = 1;
    const int cgt = ct % p.n_cgt, cxt = ct / p.n_cgt;
    const int ky0 = kg * 2;
    const bool has_ky = ky0 + z < p.k;                        // (the last "pair" of an odd window has one filter row)
    const int cg0 = cgt * 64, cx0 = cxt * 64;
    const int nr = p.nr, TWp = p.TWp, RP = p.RP, nks = p.nks;
    const int ROWB = RP * 128;

    // X offset of run u (8 pixels; lane half h of k-step t is run 2t + h) inside an X image of nr + 1 rows
    int* const tabx = reinterpret_cast<int*>(sm + WR_TAB);
    if (!RING && tid < 32) {
        const int px = tid * 8, rowin = px / TWp, col = px - rowin * TWp;
        tabx[tid] = (tid < 2 * nks && rowin < nr) ? rowin * ROWB + col * 128 : 0;   // (a run past the stage pairs a zero dY with finite X)
    }

    // Raw buffer descriptors {base, stride 0, bytes, flags}, kept as plain SGPR quads: the LDS-DMA instructions are issued as
    // inline asm.  Through the builtin the compiler sees an LDS store it cannot tell apart from the addresses of the
    // ds_read_b64_tr_b16 builtins and puts s_waitcnt vmcnt(0) in front of every fragment read that follows a DMA -- the whole
    // prefetch would be waited for at the first k-step of every stage.  The kernel orders DMA and reads itself (vmcnt(0) + barrier
    // at the stage boundary); nothing else in it uses M0.
    auto make_rsrc = [](const void* ptr, unsigned bytes) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
        return wr_i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                        __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
    };
    const wr_i32x4 rs_g = make_rsrc(p.g, p.g_bytes), rs_x = make_rsrc(p.x, p.x_bytes);
    const unsigned sm_base = (unsigned)(size_t)(lds_u8*)sm;
    const unsigned OOB = 0xFFFFFF00u;
    const int prow = lane >> 3;                                                  // row of a DMA piece this lane fills
    const unsigned lc16 = (unsigned)((lane & 7) ^ (((prow >> 1) & 1) << 2)) * 16u;   // logical chunk behind its physical slot

    // ---- the LDS-DMA pieces of a stage (1 KB = 8 pixels x 64 channels each; piece e of a wave = piece e * 8 + wave of the image)
    // are FIRED one per k-step under the MFMAs of the stage before (eight DMA instructions in a row cost a wave 500-1400 cycles of
    // issue with nothing else going on), addresses computed on the spot from a handful of stage scalars.  Pieces 0 .. NXE-1 of a
    // wave: X; NXE .. NP-1: dY.  A piece that does not exist (past the image, the dY of a virtual stage, anything after the
    // last stage) is an out-of-range read into a 1 KB dummy slot: every wave fires NP pieces per stage, whatever the stage.
    constexpr int NGE = WR_GPX / 64 + 1;                       // dY pieces per wave (28 pieces of 8 pixels over 8 waves)
    constexpr int NXE = RING ? (WR_GPX + 16) / 64 + 1 : 6;     // X pieces per wave: one row of <= 30 pieces, or <= 44 over the image
    constexpr int NP = NXE + NGE;
    const int nxr = RP >> 3;                                   // pieces per X row
    const int nxp = RING ? nxr : (nr + 1) * nxr;
    int pf_b = 0, pf_x0 = 0, pf_oy0 = 0, pf_c = 0;             // the stage being fetched: image, first column, first row, counter
    bool pf_virt = false, pf_live = false;
    // (branch-free, so that a k-step stays ONE basic block the scheduler can interleave: even k-steps fire X pieces, odd ones dY)
    const bool refl = p.pad_mode == 1;
    auto fire_x = [&](int e) {
        const int pe = e * 8 + wave;
        const bool pok = pf_live & (e < NXE) & (pe < nxp);                    // (bitwise: no short-circuit branches)
        const int j = RING ? 1 : (pe * p.xmagic) >> 16;                       // = pe / nxr
        const int pc = RING ? pe : pe - j * nxr;
        const int iy0 = pf_oy0 + j + ky0 - p.pad, iy1 = reflect_idx(iy0, p.Hx);
        const int iy = refl ? iy1 : iy0;
        const bool rok = pok & ((unsigned)iy < (unsigned)p.Hx);
        const int ix0 = pf_x0 - p.pad + pc * 8 + prow, ix1 = reflect_idx(ix0, p.Wx);
        const int ix = refl ? ix1 : ix0;
        const bool ok = rok & ((unsigned)ix < (unsigned)p.Wx);
        unsigned vo_ = (unsigned)(ix * p.ldx + cx0) * 2u + lc16;
        unsigned so_ = (unsigned)(((pf_b * p.Hx + iy) * p.Wx) * p.ldx) * 2u;
        const int slot = RING ? (pf_c + 1) % 3 : (pf_c & 1) * (nr + 1) + j;
        unsigned la_ = (unsigned)(WR_X0 + slot * ROWB + pc * 1024);
        asm("" : "+v"(vo_), "+s"(so_), "+s"(la_));             // (computed unconditionally: no branch inside a k-step)
        const unsigned vo = ok ? vo_ : OOB;
        const unsigned so = rok ? so_ : 0u;
        const unsigned m0v = sm_base + (pok ? la_ : (unsigned)WR_DUMMY);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane((int)m0v)), "v"(vo), "s"(rs_x), "s"(__builtin_amdgcn_readfirstlane((int)so)) : "memory");
    };
    auto fire_g = [&](int e) {
        const int pe = e * 8 + wave;
        const bool pok = pf_live & !pf_virt & (e < NGE) & (pe < 2 * nks);
        const int px = pe * 8 + prow;
        const int rowin = RING ? 0 : (px * p.gmagic) >> 16;                   // = px / TWp
        const int col = px - rowin * TWp;
        const bool ok = pok & (rowin < nr) & (pf_oy0 + rowin < p.Hg) & (col < p.TW) & (pf_x0 + col < p.Wg);
        unsigned vo_ = (unsigned)((rowin * p.Wg + col) * p.ldg + cg0) * 2u + lc16;
        unsigned so_ = (unsigned)(((pf_b * p.Hg + pf_oy0) * p.Wg + pf_x0) * p.ldg) * 2u;
        unsigned la_ = (unsigned)((pf_c & 1) * WR_GBYTES + pe * 1024);
        asm("" : "+v"(vo_), "+s"(so_), "+s"(la_));
        const unsigned vo = ok ? vo_ : OOB;
        const unsigned so = pok ? so_ : 0u;
        const unsigned m0v = sm_base + (pok ? la_ : (unsigned)WR_DUMMY);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane((int)m0v)), "v"(vo), "s"(rs_g), "s"(__builtin_amdgcn_readfirstlane((int)so)) : "memory");
    };
    constexpr int NFS = 2 * (NXE > NGE ? NXE : NGE);           // k-steps of a stage that fire a real piece

    f32x16 acc[KW];

    // per-lane parts of the fragment addresses: pixel row q of the 4 a 16-lane group transposes, the 32-channel half of this
    // wave (swapped on pixels with bit 1 set), 8 bytes = 4 channels per lane; the second X window starts two positions on
    const unsigned offA = (unsigned)(q * 128 + ((wr ^ (q >> 1)) << 6) + 32 * g16 + 8 * pp + h * 1024);
    const unsigned offB = (unsigned)(q * 128 + ((wc ^ (q >> 1)) << 6) + 32 * g16 + 8 * pp);
    const unsigned offB2 = (unsigned)((q + 2) * 128 + ((wc ^ (((q + 2) >> 1) & 1)) << 6) + 32 * g16 + 8 * pp);

    // Two register sets: at the top of k-step t every fragment of k-step t + 1 is requested (2 dY reads, NR1 + NR2 X reads) into
    // the other set, then the KW MFMAs of step t run on the set that was requested a whole k-step earlier.  A window is ONE
    // 8-register vector: the operand of tap kx is a 4-register sub-tuple of it at an even offset (or of its shifted copy), so no
    // register is moved to form an operand.  Reads past the last k-step fetch LDS bytes nobody uses.
    s16x4 fa[2][2];
    wr_i32x8 fw[2], fw2[2];
    unsigned gbase = 0, xbase = 0;
    auto rd_all = [&](int set, int t) {
        const lds_u8* ga = (const lds_u8*)sm + gbase + t * 2048;
        fa[set][0] = tr_read(ga);
        fa[set][1] = tr_read(ga + 512);
        const unsigned xo = RING ? (unsigned)(t * 2048 + h * 1024) : (unsigned)tabx[2 * t + h];
        const lds_u8* xb = (const lds_u8*)sm + xbase + xo;
#pragma unroll
        for (int i = 0; i < NR1; ++i) {
            const int2 v = __builtin_bit_cast(int2, tr_read(xb + offB + i * 512));
            fw[set][2 * i] = v.x; fw[set][2 * i + 1] = v.y;
        }
#pragma unroll
        for (int i = 0; i < NR2; ++i) {
            const int2 v = __builtin_bit_cast(int2, tr_read(xb + offB2 + i * 512));
            fw2[set][2 * i] = v.x; fw2[set][2 * i + 1] = v.y;
        }
    };
    auto taps = [&](const wb_bf16x8 a, wr_i32x8 w, int which) {                  // the MFMAs of the taps served by one window
        // (opaque: the window becomes ONE 8-register tuple here, operands are sub-tuples of it; and the registers of elements no
        //  tap uses stay reserved until their read has landed -- reused earlier, they force a wait for the whole read-ahead)
        asm("" : "+v"(w));
        wr_i32x8 s;                                                              // the window shifted by one element
#pragma unroll
        for (int i = 0; i < 7; ++i) s[i] = (int)__builtin_amdgcn_alignbit((unsigned)w[i + 1], (unsigned)w[i], 16);
        s[7] = 0;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            if (((kx & 3) >> 1) != which) continue;
            const int a0 = 2 * (kx >> 2);
            const wr_i32x8 src = (kx & 1) ? s : w;
            const wr_i32x4 bv = a0 == 0 ? __builtin_shufflevector(src, src, 0, 1, 2, 3)
                              : a0 == 2 ? __builtin_shufflevector(src, src, 2, 3, 4, 5) : __builtin_shufflevector(src, src, 4, 5, 6, 7);
            acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(wb_bf16x8, bv), acc[kx], 0, 0, 0);
        }
    };
    // One k-step = one basic block: the 2 + NR1 + NR2 reads of step t + 1, one LDS-DMA piece with its address arithmetic and the
    // KW MFMAs of step t with their shifts, interleaved by the scheduler as "one MFMA, one or two reads, a few vector / scalar
    // instructions": everything that is not an MFMA is issued in the shadow of one.
    auto mfmas = [&](int set) {
        const wb_bf16x8 a = __builtin_bit_cast(wb_bf16x8, __builtin_shufflevector(fa[set][0], fa[set][1], 0, 1, 2, 3, 4, 5, 6, 7));
        taps(a, fw[set], 0);
        taps(a, fw2[set], 1);
    };
    auto kstep = [&](int set, int t) {
        rd_all(set ^ 1, t + 1);
        if (set == 0) fire_x(t >> 1); else fire_g(t >> 1);
        mfmas(set);
        WrSched<KW, 2 + NR1 + NR2, 0>::apply();
        __builtin_amdgcn_sched_barrier(0);
    };

    __syncthreads();                                           // the run table (no LDS-DMA in flight yet: a plain barrier)
    for (int si = 0; si < nsplit; ++si) {
        const int split = split0 + si;
        const int i0 = split * p.ips, i1 = min(p.nst, i0 + p.ips);
#pragma unroll
        for (int t = 0; t < KW; ++t) {
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) acc[t][r_] = 0.f;
        }
        // ---- stage walk: the stage computed at counter c had its DMA fired during stage c - 1; a ring starts every strip
        // column (and every split) with a virtual stage that brings in X row 0 of the filter-row pair.  The stage boundary
        // (vmcnt(0), barrier) sits INSIDE the last k-step of a stage: behind the barrier the first fragment reads of the next stage
        // go out, then the MFMAs of the last step run on operands that are in registers already -- the reads land under them
        // instead of in front of an idle matrix pipe. ----
        int rb = i0 % p.sph, strip, b;
        { const int t2 = i0 / p.sph; strip = t2 % p.nstrip; b = t2 / p.nstrip; }
        int cur_i = i0;
        bool cur_virt = RING;
        int c = 0;
        pf_b = b; pf_x0 = strip * p.TW; pf_oy0 = rb * nr - (cur_virt ? 1 : 0); pf_virt = cur_virt; pf_live = true; pf_c = 0;
        if (si > 0) __builtin_amdgcn_s_barrier();              // (every wave is done with the previous split's LDS images)
#pragma unroll 1
        for (int e = 0; e < NFS / 2; ++e) { fire_x(e); fire_g(e); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the first stage's pieces have landed
        __builtin_amdgcn_s_barrier();
        gbase = offA;
        xbase = (unsigned)(WR_X0 + (RING ? (z % 3) * ROWB : z * ROWB));
        rd_all(0, 0);
        for (;;) {
            bool nxt_virt = false, has_next = true;
            if (!cur_virt) {
                has_next = cur_i + 1 < i1;
                if (has_next) {
                    ++cur_i;
                    if (++rb == p.sph) { rb = 0; if (++strip == p.nstrip) { strip = 0; ++b; } }
                    nxt_virt = RING && rb == 0;
                }
            }
            pf_b = b; pf_x0 = strip * p.TW; pf_oy0 = rb * nr - (nxt_virt ? 1 : 0); pf_virt = nxt_virt; pf_c = c + 1;
            pf_live = has_next && !(p.knobs & 1);              // (timing knob 1: the loop without its DMA traffic)
            // (no branch around the stage's MFMA loop: a stage this wave does not compute -- virtual, or the missing filter row of
            //  an odd window -- is a loop of zero k-steps; the register allocator otherwise keeps a second copy of the accumulators
            //  for the path around it)
            const int nk = (!cur_virt && has_ky && !(p.knobs & 2)) ? nks : 0;
            // a stage with an odd number of k-steps keeps its last one (set 0) for the boundary; an even one ends on set 1 and the
            // boundary is bare (the layers of the two networks all have 13 k-steps per stage)
            const int nlast = nk & 1, npair = nk - nlast;
            int t = 0;
#pragma nounroll
            for (; t < npair; t += 2) { kstep(0, t); kstep(1, t + 1); }
#pragma unroll 1
            for (int u = t; u < NFS; ++u) { if (u & 1) fire_g(u >> 1); else fire_x(u >> 1); }   // (pieces the k-steps did not fire)
            // ---- the stage boundary, in front of the last k-step's MFMAs (timing knob 4: behind them, the bare boundary) ----
            const int npre = (p.knobs & 4) ? nlast : 0;
#pragma nounroll
            for (int i = 0; i < npre; ++i) mfmas(0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // the next stage's pieces have landed; this wave's reads of this stage too
                __builtin_amdgcn_s_barrier();                   // ... everyone's
                gbase = (unsigned)(((c + 1) & 1) * WR_GBYTES) + offA;
                xbase = (unsigned)(WR_X0 + (RING ? ((c + 1 + z) % 3) * ROWB : (((c + 1) & 1) * (nr + 1) + z) * ROWB));
            }
            rd_all(1, 0);                                      // the next stage's first fragments ...
            __builtin_amdgcn_sched_barrier(0);
#pragma nounroll
            for (int i = 0; i < nlast - npre; ++i) mfmas(0);   // (zero or one trip: a loop, not a branch -- see nk above)
            __builtin_amdgcn_sched_barrier(0);
            fa[0][0] = fa[1][0]; fa[0][1] = fa[1][1]; fw[0] = fw[1]; fw2[0] = fw2[1];      // ... over to set 0, where a stage starts
            if (!has_next) break;
            cur_virt = nxt_virt; ++c;
        }

        // ---- epilogue: slab [split][ky * k + kx][Cg][Cx]; wave (z, wr, wc) holds filter row ky0 + z, dY channels wr * 32 ..,
        // X channels wc * 32 ..
        if (has_ky) {
            const int KK = p.k * p.k;
            const int cx = cx0 + wc * 32 + (lane & 31);
#pragma unroll
            for (int t = 0; t < KW; ++t) {
                float* dst = p.part + (((size_t)split * KK + (size_t)(ky0 + z) * p.k + t) * p.Cg + cg0 + wr * 32 + 4 * h) * p.Cx + cx;
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) dst[(size_t)((r_ & 3) + 8 * (r_ >> 2)) * p.Cx] = acc[t][r_];
            }
        }
    }
#endif  // WR_DEVICE_BODY
}
