// conv_ring2_bf16: the LDS-DMA ring kernel (conv_ring.h) with ONE barrier per (filter row, slab) stage (round 5).
// (included by conv_igemm.hip after conv_ring.h)
//
// conv_ring_bf16 synchronises its eight waves at every tap step -- 8 MFMAs per wave between two raw barriers, with two waves
// per SIMD and nobody else on the CU: 45-55 % of the matrix pipe's cycles inside the tap loop go to barrier skew and counted
// vmcnt waits (r04b_ring_probe.txt; NOT to LDS-DMA bytes: a third less traffic bought 3 %).  wgrad_ring_bf16 (wgrad_ring.h),
// which synchronises once per stage, keeps the pipe 67 % busy.  This kernel gives the forward / data-gradient the same shape:
//   * a stage is (filter row, 32-CHANNEL slab): LDS rows are 64 bytes; the patch of a 512-pixel tile with its halos is 40 KB
//     (640 positions) and the KW weight tiles of the stage are KW x 4 KB (BN = 64) -- small enough to hold TWO whole stages
//     (patch + every weight tile: 2 x 76 KB for a 9-tap row) in LDS.  While stage s computes, every wave fires its ten pieces
//     of stage s + 1 into the other pair, one or two per tap step under the MFMAs; the boundary (vmcnt(0) lgkmcnt(0), barrier)
//     sits inside the last k-substep in front of its MFMAs, and the first fragment reads of the next stage land under them.
//     72 MFMAs per wave run between two barriers.
//   * 512-pixel tiles: the 8 waves are 8 pixel groups of 64 (no split of the reduction between waves, no exchange in the
//     epilogue), every weight tile serves twice the pixels of conv_ring_bf16's, and the ~7 us a tile spends outside its tap loop
//     are paid half as often;
//   * 64-byte rows: the four 16-byte chunks of a row are XOR-permuted by (row >> 2) & 3, so the 16 rows a ds_read_b128 lane
//     group touches fall on 16 different 16-byte bank slots; an LDS-DMA piece is 16 rows; the permutation is applied to the
//     per-lane SOURCE address and on the read;
//   * everything else -- persistent workgroups, tables, the next tile's prologue before the epilogue, the K-split tail, the
//     epilogue fusions -- is conv_ring_bf16's.
// One-stage-ahead prefetch needs a stage longer than the DMA latency: 9 / 7-tap rows (2.2 / 1.7 us of MFMA work per stage); the
// 5- and 3-tap layers stay on conv_ring_bf16's four-step-ahead ring.  B = 20, 9x9 64 -> 64: 1249 vs 1155 TFLOP/s.
#pragma once

#define RG2_BM 512
#define RG2_APOS 640                      // staged patch positions per A buffer: 5 LDS-DMA pieces (16 positions each) per wave
#define RG2_ABYTES (RG2_APOS * 64)
#define RG2_NRMAX 12                      // image rows a 512-pixel tile may touch (W >= 52)

// BNB: the data-gradient instantiation whose epilogue also emits the producer BatchNorm's backward partials (p.bnb_y; kept out of
// the other instantiations: its 16 running sums cost the 256 x 128 form ~30 spilled registers)
template <int BN, int KW, int DPO = 0, bool BNB = false>
__global__ __launch_bounds__(512, 2) void conv_ring2_bf16(const IgemmParams p) {
#if RG_DEVICE_BODY
    constexpr int BM = RG2_BM;
    constexpr int TILE_B = BN * 64;                          // weight tile: BN rows of 32 channels
    constexpr int PT = BN / 16;                              // LDS-DMA pieces per weight tile (16 rows x 64 bytes each)
    constexpr int NBW = (KW * PT + 7) / 8;                   // weight pieces per wave and stage (all KW tiles of the stage)
    constexpr int NG = 2;                                    // 16-channel k-substeps per step
    constexpr int NJ = BN / 32;                              // column tiles per wave
    static_assert(PT == 4 || PT == 8, "a wave's pieces all cover the same rows of their tiles");
    // LDS map: [A buffer 0 | weight set 0 | A buffer 1 | weight set 1 | two table sets | coefficients].  A stage reads one
    // (A buffer, weight set) pair while the LDS-DMA of the next stage fills the other.  The next tile's prologue lands in pair
    // 0; pair 1 is the epilogue's scratch meanwhile.
    constexpr int A0 = 0, B0 = RG2_ABYTES, A1 = B0 + KW * TILE_B, B1 = A1 + RG2_ABYTES, TAB0 = B1 + KW * TILE_B;
    constexpr int SC0 = A1;                                  // epilogue scratch: [SC0, TAB0)
    constexpr int TABN = BM + RG_KMAX * RG2_NRMAX + RG2_NRMAX + 1 + RG2_NRMAX + 3;     // ints per table set (648)
    constexpr int TABSET = TABN * 4;
    static_assert(TAB0 - SC0 >= 35 * 1024, "LDS map");
    // ONE shared object: the compiler must see a single LDS array beside the LDS-DMA instructions
    constexpr int COEF0 = TAB0 + 2 * TABSET + 64;            // [4][BN] floats: the BatchNorm coefficients of a data gradient's bnb mode
    static_assert(COEF0 + 4 * 128 * 4 <= 160 * 1024, "LDS size");
    __shared__ __attribute__((aligned(16))) unsigned char sm[COEF0 + 4 * 128 * 4];
    int* const wirow = reinterpret_cast<int*>(sm + TAB0 + 2 * TABSET);    // [RG_KMAX] weight index of the first tap of each filter row
    auto tab = [&](int b) { return reinterpret_cast<int*>(sm + TAB0 + b * TABSET); };
    // a table set: row_out[512] output pixel index or -1 | rowoff[RG_KMAX][RG2_NRMAX] byte offset of input row or -1 |
    //              rbase[RG2_NRMAX + 1] first patch position of each touched image row | rxlo[RG2_NRMAX] first output column of
    //              the tile in that row | {nrows, b_first}
    constexpr int T_ROWOFF = BM, T_RBASE = T_ROWOFF + RG_KMAX * RG2_NRMAX, T_RXLO = T_RBASE + RG2_NRMAX + 1, T_MISC = T_RXLO + RG2_NRMAX;

    const int knobs = p.kc;                                  // measurement knobs (0 in production)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;                                     // pixel group: tile rows [64 wm, 64 wm + 64)
    const IgemmPhase ph = p.ph[0];
    const int Wo = ph.Wo, Ho = ph.Ho;
    const int M = p.B * Ho * Wo;
    const int ntap = ph.tap_end - ph.tap_begin;
    const int kh = ntap / KW;
    // taps of a filter row are consecutive in the weight tensor and in dx (ring_kw() checked it): ascending dx for a forward
    // convolution, descending for a data gradient -- no per-tap table is read inside the loop (a kernarg table indexed at run
    // time becomes a vector-memory load, whose wait would drain the LDS-DMA pipeline)
    const int dx_first = p.tdx[ph.tap_begin], dx_last = p.tdx[ph.tap_begin + KW - 1];
    const int dxmin = min(dx_first, dx_last);
    const bool dx_up = dx_last >= dx_first;
    const int nchunks = p.Cred / 32;
    // (filter row, channel slab) stages; knob bit 0: skip the loop, bit 3: a third of it (what a tile costs outside / per stage)
    const int nstage = (knobs & 1) ? 0 : (knobs & 8) ? kh * nchunks / 3 : kh * nchunks;
    const int tap2 = p.w_tap_stride * 2;

    // ---- this workgroup's work items.  Workgroups b, b + 8, ... share an XCD (its L2): an XCD owns a contiguous band of units
    // ((M-tile, N-tile), N-tiles adjacent; neighbouring tiles re-read each other's halo rows), walked by its workgroups side by
    // side.  B = 20 gives 65/64 of a power of two tiles at every level of the pyramid: the last, nearly empty round would cost
    // a whole tile time.  So when the host planned a split (p.ksplit = parts > 1), the p.ring_main units of the full rounds are
    // walked as bands and each unit of the last round is cut into `parts` ranges of p.ring_sp stages, one per workgroup, whose
    // raw fp32 accumulators go to slabs in p.part; splitk_combine_kernel sums them and applies the epilogue for those pixels.
    const int bid = blockIdx.x, xcd = bid & 7, wg_in_xcd = bid >> 3, wgs_per_xcd = (int)gridDim.x >> 3;
    const int parts = (knobs & 9) ? 1 : p.ksplit;
    const int per_xcd = (p.grid_m + 7) >> 3;
    const int units_xcd = parts > 1 ? p.ring_main >> 3 : per_xcd * p.grid_n;
    const int rounds_main = parts > 1 ? units_xcd / wgs_per_xcd : (units_xcd + wgs_per_xcd - 1) / wgs_per_xcd;
    const int tail_items = parts > 1 ? (p.grid_m * p.grid_n - p.ring_main) * parts : 0;
    // item k of this workgroup -> (M-tile, N-tile, stage range, slab index or -1); false: no such item
    auto get_item = [&](int k, int& mt, int& nt, int& s0, int& s1, int& part) {
        if (k < rounds_main) {
            const int q = wg_in_xcd + k * wgs_per_xcd;
            if (q >= units_xcd) return false;
            const int u = parts > 1 ? xcd * units_xcd + q : (xcd * per_xcd + q / p.grid_n) * p.grid_n + q % p.grid_n;
            mt = u / p.grid_n; nt = u % p.grid_n; s0 = 0; s1 = nstage; part = -1;
            return mt < p.grid_m;
        }
        if (k == rounds_main && bid < tail_items) {
            const int u = p.ring_main + bid / parts;
            part = bid % parts;
            mt = u / p.grid_n; nt = u % p.grid_n;
            s0 = part * p.ring_sp; s1 = min(nstage, s0 + p.ring_sp);
            return s0 < s1;
        }
        return false;
    };

    // tables of the tile with first pixel m0 into set b (every thread takes part; the caller orders them with a barrier)
    auto setup_tables = [&](int b, int m0) {
        int* t = tab(b);
        const int mlast = min(m0 + BM, M) - 1;
        const int row0 = m0 / Wo, nrows = mlast / Wo - row0 + 1;
        const int b_first = row0 / Ho;
        {
            const int m = m0 + tid;
            int v = -1;
            if (m < M) {
                const int ox = m % Wo, tt = m / Wo, oy = tt % Ho, b_ = tt / Ho;
                v = (b_ * p.Hy + oy * p.osy + ph.oy0) * p.Wy + ox * p.osx + ph.ox0;
            }
            t[tid] = v;
        }
        if (tid <= nrows) {
            const int xlo0 = m0 - row0 * Wo;
            const int before = tid == 0 ? 0 : (Wo - xlo0) + (tid - 1) * Wo;
            t[T_RBASE + tid] = (tid == nrows ? min(BM, M - m0) : before) + tid * (KW - 1);
            if (tid < nrows) t[T_RXLO + tid] = tid == 0 ? xlo0 : 0;
        }
        if (tid == 511) { t[T_MISC] = nrows; t[T_MISC + 1] = b_first; }
        if (tid >= 256 && tid < 256 + kh * RG2_NRMAX) {
            const int i = tid - 256;
            const int ky = i / RG2_NRMAX, j = i - ky * RG2_NRMAX;
            int off = -1;
            if (j < nrows) {
                const int tt = row0 + j, oy = tt % Ho, b_ = tt / Ho;
                int iy = oy * p.stride + p.tdy[ph.tap_begin + ky * KW];
                if (p.pad_mode == 1) iy = reflect_idx(iy, p.Hi);
                if ((unsigned)iy < (unsigned)p.Hi) off = (((b_ - b_first) * p.Hi + iy) * p.Wi) * p.ldx1 * 2;
            }
            t[T_ROWOFF + i] = off;
        }
    };

    // ---- per-tile state ----
    const unsigned OOB = 0xFFFFFF00u;
    constexpr int NA = RG_AV, NB = 1;
    unsigned pk[NA];             // patch piece e of this wave = piece e * 8 + wave of the stage's 40 (16 positions x 4 chunks): per-lane
                                 // column offset, with the image-row index j of the position in its low 4 bits (the offset is a multiple of 16)
    unsigned boff[NB];           // this wave's weight pieces are rows [16 (wave % PT), + 16) of their tiles (16 rows x 4 chunks)
    unsigned a_vo[NA];           // per-lane source offsets of the patch pieces of one stage
    __amdgpu_buffer_rsrc_t rs_x;
    // (p.kc carries measurement knobs for this kernel: bit 1 / bit 2 give the weight / activation descriptor zero records, so
    //  the range check drops every LDS-DMA through it while the instruction stream, the waits and the barriers stay: what the
    //  loop costs without that operand's traffic.  Timing only -- the results are wrong.)
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (knobs & 2) ? 0 : (int)p.w_bytes, 0x00020000);

    auto tile_dma_state = [&](int b, int n0) {      // pk / boff / the activation descriptor of the tile whose tables are set b
        const int* t = tab(b);
        const int nrows = __builtin_amdgcn_readfirstlane(t[T_MISC]), b_first = __builtin_amdgcn_readfirstlane(t[T_MISC + 1]);
        const int npatch = t[T_RBASE + nrows];
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int q = (e * 8 + wave) * 16 + (lane >> 2);
            int j = 0;
            for (int jj = 1; jj < nrows; ++jj) j += (q >= t[T_RBASE + jj]) ? 1 : 0;
            int ix = t[T_RXLO + j] + dxmin + (q - t[T_RBASE + j]);
            if (p.pad_mode == 1) ix = reflect_idx(ix, p.Wi);
            const bool ok = q < npatch && (unsigned)ix < (unsigned)p.Wi;
            const unsigned lc = (unsigned)((lane & 3) ^ ((q >> 2) & 3));              // logical chunk stored at this physical slot
            pk[e] = ok ? ((unsigned)(ix * p.ldx1) * 2u + lc * 16u) | (unsigned)j : OOB;
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int n = (wave & (PT - 1)) * 16 + (lane >> 2);
            const unsigned lc = (unsigned)((lane & 3) ^ ((n >> 2) & 3));
            boff[e] = (n < BN && (n0 + n) < p.N) ? (unsigned)((n0 + n) * p.Cred) * 2u + lc * 16u : OOB;
        }
        // (descriptor words through readfirstlane: a descriptor the compiler cannot PROVE wave-uniform gets a waterfall loop
        //  around every LDS-DMA instruction)
        const unsigned long long img1 = (unsigned long long)p.Hi * p.Wi * p.ldx1 * 2ull * b_first;
        const unsigned long long rem1 = p.x_bytes - img1, cap = 0xFF000000ull;
        const unsigned long long xb = reinterpret_cast<unsigned long long>(p.x) + img1;
        const unsigned long long xbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xb >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xb);
        const int xlen = __builtin_amdgcn_readfirstlane((knobs & 4) ? 0 : (int)(unsigned)(rem1 < cap ? rem1 : cap));
        rs_x = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(xbu), 0, xlen, 0x00020000);
    };
    auto a_offsets = [&](int b, int ky) {                    // (read from the row table once per stage, not under the MFMAs)
        const int* t = tab(b);
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int ro = t[T_ROWOFF + ky * RG2_NRMAX + (int)(pk[e] & 15u)];
            a_vo[e] = (ro >= 0 && pk[e] != OOB) ? (unsigned)ro + (pk[e] & ~15u) : OOB;
        }
    };
    auto dma_a = [&](int par, int cc, int e) {               // piece e of the patch whose offsets are in a_vo -> A buffer `par`
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (rg_lds_ptr)(sm + (par ? A1 : A0) + (e * 8 + wave) * 1024), 16, a_vo[e], cc * 64, 0, 0);
    };
    // piece e of this wave of the KW weight tiles of the stage whose first tap lies at scalar offset soff -> weight set `par`:
    // piece e * 8 + wave of the KW * PT (tile kx = piece / PT at soff + kx * tap2).  A piece past the last one re-loads the wave's
    // previous piece (same bytes to the same place), so that every wave issues the same number of loads.
    auto dma_bw = [&](int par, int soff, int e) {
        int pi = e * 8 + wave;
        if (pi >= KW * PT) pi -= 8;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (rg_lds_ptr)(sm + (par ? B1 : B0) + pi * 1024), 16, boff[0],
                                                 soff + (pi / PT) * tap2, 0, 0);
    };
    // The weights of tap kx of a stage lie at stage_soff + kx * tap2; past the last stage the "next" stage is the last stage
    // again: harmless reloads into the idle pair keep the instruction stream uniform.
    auto stage_soff = [&](int ky_, int cc_) { return __builtin_amdgcn_readfirstlane(wirow[ky_]) * tap2 + cc_ * 64; };
    int ky_n = 0, cc_n = 0, soff_c = 0, soff_n = 0;
    // (measured and not kept: every workgroup starting at another filter row, so that the workgroups of an XCD read different
    //  weight tiles at any moment: 1049 vs 1100 TFLOP/s -- sharing the lines helps)
    auto issue_prologue = [&](int b, int s0, int ns) {       // first stage's patch and weight tiles of the item in table set b -> pair 0
        const int ky0 = s0 / nchunks, cc0 = s0 - ky0 * nchunks;
        ky_n = ky0; cc_n = cc0;
        if (ns > 1) { if (++cc_n == nchunks) { cc_n = 0; ++ky_n; } }
        soff_c = stage_soff(ky0, cc0); soff_n = stage_soff(ky_n, cc_n);
        a_offsets(b, ky0);
#pragma unroll
        for (int e = 0; e < NA; ++e) dma_a(0, cc0, e);
#pragma unroll
        for (int e = 0; e < NBW; ++e) dma_bw(0, soff_c, e);
    };

    // ---- fragment addressing ----
    // k-substep g of a step covers channels [16 g, 16 g + 16) of the slab: lane half h reads chunk 2 g + h
    const unsigned hbit = (unsigned)(lane >> 5) << 4;
    unsigned bfix[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const unsigned n = (unsigned)(j * 32 + (lane & 31));
        bfix[j] = ((n << 6) ^ (((n >> 2) & 3u) << 4)) ^ hbit;
    }
    unsigned apos[2];                                // A: pixel r of the tile sits at patch position r + j(r) * (KW - 1) (+ the tap's shift)
    f32x16 acc[2][NJ];
    bf16x8 fa[2][2], fb[2][NJ];                      // [register set][row tile / column tile]
    auto load_frags = [&](int set, int par, int kx, int g) {     // fragments of (pair par, tap kx, k-substep g)
        const unsigned sh = (unsigned)(dx_up ? kx : KW - 1 - kx);
        const unsigned abase = (unsigned)(par ? A1 : A0);
        const unsigned bbase = (unsigned)((par ? B1 : B0) + kx * TILE_B);
        const unsigned gx = (unsigned)g << 5;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned q = apos[i] + sh;
            const unsigned a = abase + (((q << 6) ^ (((q >> 2) & 3u) << 4)) ^ hbit ^ gx);
            fa[set][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(sm + a));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            fb[set][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(sm + bbase + (bfix[j] ^ gx)));
    };

    // ---- first item ----
    int k_item = 0, mt = 0, nt = 0, s0 = 0, s1 = 0, part = -1;
    if (!get_item(0, mt, nt, s0, s1, part)) {
        k_item = rounds_main;                        // (no main unit for this workgroup: it may still own a tail range)
        if (!get_item(k_item, mt, nt, s0, s1, part)) return;
    }
    if (tid < kh) wirow[tid] = p.twi[ph.tap_begin + tid * KW];
    setup_tables(0, mt * BM);
    __syncthreads();
    tile_dma_state(0, nt * BN);
    issue_prologue(0, s0, s1 - s0);
    int tb = 0;

    for (;;) {
        const int m0 = mt * BM, n0 = nt * BN, ns = s1 - s0;
        int mt_n = 0, nt_n = 0, s0_n = 0, s1_n = 0, part_n = -1, k_n = k_item + 1;
        bool has_next = get_item(k_n, mt_n, nt_n, s0_n, s1_n, part_n);
        if (!has_next && k_n < rounds_main) { k_n = rounds_main; has_next = get_item(k_n, mt_n, nt_n, s0_n, s1_n, part_n); }
        const int* const t = tab(tb);
        {
            const int mlast = min(m0 + BM, M) - 1;
            const int row0 = m0 / Wo;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + (lane & 31);
                const int m = min(m0 + r, mlast);
                apos[i] = (unsigned)(min(r, mlast - m0) + (m / Wo - row0) * (KW - 1));
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (has_next) setup_tables(tb ^ 1, mt_n * BM);              // (the other set: ordered by the loop's barriers before anyone reads it)
        rg_wait_vm<0>();                                            // the prologue pieces (and the previous epilogue's stores)
        __builtin_amdgcn_s_barrier();
        if (ns > 0) load_frags(0, 0, 0, 0);

        // ---- stage walk: ONE barrier per stage.  During stage s every wave fires its pieces of stage s + 1 (patch and all KW
        // weight tiles) into the other pair, one or two per tap step under the MFMAs; the boundary -- vmcnt(0) lgkmcnt(0), barrier --
        // sits inside the last k-substep, in front of its MFMAs, whose operands are in registers by then: behind the barrier the
        // first fragments of the next stage are requested and land under those MFMAs.  Between two boundaries a wave runs
        // 2 KW NJ MFMAs (72 for a 9-tap row) with no synchronisation at all.
        for (int stage = 0; stage < ns; ++stage) {         // (stage: local index -- the pair parity starts at 0 for every item)
            const int par = stage & 1;
            a_offsets(tb, ky_n);                               // (ky_n, cc_n): the NEXT stage; the last stage reloads itself into the idle pair
            const bool more = stage + 1 < ns;
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int cur = g & 1;
                    if (g + 1 < NG) load_frags(cur ^ 1, par, kx, g + 1);
                    else if (kx + 1 < KW) load_frags(cur ^ 1, par, kx + 1, 0);
                    else {
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) {
                            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next stage has landed; this wave's reads of this one too
                            __builtin_amdgcn_s_barrier();                              // ... everyone's
                            load_frags(cur ^ 1, par ^ 1, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
                    if (g == 0) {
                        // this step's share of the next stage's LDS-DMA, under the MFMAs: patch piece e at step e (KW - 1) / NA,
                        // weight piece e at step e (KW - 1) / NBW -- nothing in the last step, whose boundary waits for them
#pragma unroll
                        for (int e = 0; e < NA; ++e)
                            if (e * (KW - 1) / NA == kx) dma_a(par ^ 1, cc_n, e);
#pragma unroll
                        for (int e = 0; e < NBW; ++e)
                            if (e * (KW - 1) / NBW == kx) dma_bw(par ^ 1, soff_n, e);
                    }
                    // (conv_ring_bf16's schedule: one MFMA, one fragment read of the NEXT k-substep, a couple of vector / scalar
                    //  instructions and at most one LDS-DMA per group)
                    if (g + 1 < NG || kx + 1 < KW) {
#pragma unroll
                        for (int i = 0; i < 2 * NJ; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            soff_c = soff_n;
            if (stage + 2 < ns) { if (++cc_n == nchunks) { cc_n = 0; ++ky_n; } }
            soff_n = stage_soff(ky_n, cc_n);
        }
        rg_wait_vm<0>();
        __syncthreads();                             // every DMA landed, every fragment read done: LDS is free; the next tile's tables are visible

        // ---- the next tile's prologue goes out before this tile's epilogue ----
        if (has_next) {
            tile_dma_state(tb ^ 1, nt_n * BN);
            issue_prologue(tb ^ 1, s0_n, s1_n - s0_n);
        }

        // ---- a tail range: the raw fp32 accumulators go to this range's slab(s); splitk_combine_kernel does the rest ----
        if (part >= 0) {
            int lane_p = lane;
            asm volatile("" : "+v"(lane_p));         // (as lane_e below: keep these addresses out of the tap loop's registers)
            const int m_tail0 = (p.ring_main / p.grid_n) * BM;
            const size_t tail_px = (size_t)(M - m_tail0);
            float* slab = p.part + ((size_t)part * tail_px + (size_t)(m0 - m_tail0)) * p.N + n0 + (lane_p & 31);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane_p >> 5);
                    if (t[row] >= 0) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) slab[(size_t)row * p.N + j * 32] = acc[i][j][r];
                    }
                }
        } else
        // ---- epilogue (scratch: [SC0, TAB0), behind the LDS the next prologue lands in) ----
        {
        // (lane_e: the epilogue's lane-dependent addresses are invariant across tiles, and hoisted out of the tile loop they would
        //  occupy ~60 registers throughout the tap loop -- an opaque copy of the lane index keeps them inside the epilogue)
        int lane_e = lane, tid_e = tid;
        asm volatile("" : "+v"(lane_e), "+v"(tid_e));
        const int col_l = lane_e & 31, rsh = 4 * (lane_e >> 5);
        float* const sc = reinterpret_cast<float*>(sm + SC0);
        constexpr int NI = 2;                        // row tiles of this wave
        // row tile ii of this wave covers tile rows rbeg(ii) + rowmap(r, lane_e)
        auto rbeg = [&](int ii) { return wm * 64 + ii * 32; };
        if (p.stats && !(BNB && p.bnb_y)) {
            float* red = sc;                         // [8 waves][BN columns][2]
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ii = 0; ii < NI; ++ii)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbeg(ii) + (r & 3) + 8 * (r >> 2) + rsh;
                        const float v = t[row] >= 0 ? acc[ii][j][r] : 0.f;
                        s1 += v; s2 += v * v;
                    }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane_e < 32) {
                    red[(wave * BN + j * 32 + lane_e) * 2 + 0] = s1;
                    red[(wave * BN + j * 32 + lane_e) * 2 + 1] = s2;
                }
            }
            __syncthreads();
            if (tid_e < BN && n0 + tid_e < p.N) {
                float s1 = 0.f, s2 = 0.f;            // (fixed order over the eight pixel groups)
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    s1 += red[(w * BN + tid_e) * 2];
                    s2 += red[(w * BN + tid_e) * 2 + 1];
                }
                p.stats[((size_t)mt * 2 + 0) * p.N + n0 + tid_e] = s1;
                p.stats[((size_t)mt * 2 + 1) * p.N + n0 + tid_e] = s2;
            }
            __syncthreads();
        }
        // output: 8192 values per round go through an fp32 LDS tile [pixel][BN + 4] and leave as whole 128-byte lines
        // (16 bytes = 8 channels per lane_e); the eval-BN affine / ReLU are applied on the way in, the residual is read 16
        // bytes at a time and added (in fp32, before the one rounding to bf16) on the way out
        const bool has_affine = p.ep_scale != nullptr;
        float es[NJ], et[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + j * 32 + col_l;
            es[j] = (has_affine && n < p.N) ? p.ep_scale[n] : 1.f;
            et[j] = (has_affine && n < p.N) ? p.ep_shift[n] : 0.f;
        }
        constexpr int RS = BN + 4;                   // row stride of the transposition tile in floats
        constexpr int PXR = 8192 / BN;               // pixels per round: 128 (BN = 64: waves 2 rr, 2 rr + 1) or 64 (BN = 128: wave rr)
        constexpr int NRND = BM / PXR;
        constexpr int TPP = BN / 8;                  // threads per pixel on the way out
        unsigned short* yo = reinterpret_cast<unsigned short*>(p.y);
        const unsigned short* ad = reinterpret_cast<const unsigned short*>(p.addsrc);
        // bnb mode (a data gradient that is the final gradient of z = [relu](BN_train(bnb_y))): the BatchNorm backward's partial
        // sums sum dz, sum dz * xhat over this tile's pixels, from the values as they are stored (rounded to bf16), so the
        // stand-alone reduce pass over (dx, y) disappears; slot = tile, like the forward's sum / sum of squares
        float* const coef = reinterpret_cast<float*>(sm + COEF0);
        const bool bnb = BNB && p.bnb_y != nullptr;
        float bs1[8], bs2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; }
        if (BNB && bnb) {
            for (int i = tid_e; i < 4 * BN; i += 512) {
                const int n = n0 + i % BN;
                coef[i] = n < p.N ? p.bnb_co[(size_t)(i / BN) * p.N + n] : 0.f;
            }
        }
#pragma unroll 1
        for (int rr = 0; rr < NRND; ++rr) {
            if ((BN == 64 ? (wm >> 1) : wm) == rr) {             // (PXR = 128: two pixel groups per round; 64: one)
#pragma unroll
                for (int ii = 0; ii < NI; ++ii)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbeg(ii) + (r & 3) + 8 * (r >> 2) + rsh - rr * PXR;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            float v = acc[ii][j][r];
                            if (has_affine) v = v * es[j] + et[j];
                            if (p.act & GDN_ACT_RELU) v = fmaxf(v, 0.f);
                            sc[row * RS + j * 32 + col_l] = v;
                        }
                    }
            }
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < PXR * TPP / 512; ++ps) {
                const int px = ps * (512 / TPP) + tid_e / TPP, cg = tid_e % TPP;
                const int op = t[rr * PXR + px];
                if (op >= 0 && n0 + cg * 8 < p.N) {
                    f32x4 lo = *reinterpret_cast<const f32x4*>(sc + px * RS + cg * 8);
                    f32x4 hi = *reinterpret_cast<const f32x4*>(sc + px * RS + cg * 8 + 4);
                    if (ad) {
                        f32x4 alo, ahi;
                        ld8_any(ad, (size_t)op * p.ld_add + n0 + cg * 8, 1, alo, ahi);
                        lo += alo; hi += ahi;
                    }
                    if (p.act & GDN_ACT_TANH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { lo[e] = tanhf(lo[e]); hi[e] = tanhf(hi[e]); }
                    }
                    st8_any(yo, (size_t)op * p.ldy + n0 + cg * 8, lo, hi, 1);
                    if (BNB && bnb) {
                        f32x4 ylo, yhi;
                        ld8_any(p.bnb_y, (size_t)op * p.ld_bnb + n0 + cg * 8, 1, ylo, yhi);
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const f32x4 yv = hh ? yhi : ylo, dv = hh ? hi : lo;
                            const f32x4 csc = *reinterpret_cast<const f32x4*>(coef + cg * 8 + hh * 4);
                            const f32x4 csh = *reinterpret_cast<const f32x4*>(coef + BN + cg * 8 + hh * 4);
                            const f32x4 cmu = *reinterpret_cast<const f32x4*>(coef + 2 * BN + cg * 8 + hh * 4);
                            const f32x4 cis = *reinterpret_cast<const f32x4*>(coef + 3 * BN + cg * 8 + hh * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float dz = bf16_h_to_f32(f32_to_bf16_h(dv[e]));
                                if (p.bnb_relu && !(yv[e] * csc[e] + csh[e] > 0.f)) dz = 0.f;
                                bs1[hh * 4 + e] += dz;
                                bs2[hh * 4 + e] += dz * ((yv[e] - cmu[e]) * cis[e]);
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (BNB && bnb) {
            // thread (pixel lane q = tid / TPP, channel group cg = tid % TPP) -> per-channel sums over the 512 / TPP pixel lanes
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[tid_e * 16 + e] = bs1[e]; sc[tid_e * 16 + 8 + e] = bs2[e]; }
            __syncthreads();
            if (tid_e < 2 * BN) {
                const int c = tid_e % BN, which = tid_e / BN;
                float a = 0.f;
                for (int q = 0; q < 512 / TPP; ++q) a += sc[(q * TPP + (c >> 3)) * 16 + which * 8 + (c & 7)];
                if (n0 + c < p.N) p.stats[((size_t)mt * 2 + which) * p.N + n0 + c] = a;
            }
            __syncthreads();
        }
        }
        if (!has_next) break;
        k_item = k_n; mt = mt_n; nt = nt_n; s0 = s0_n; s1 = s1_n; part = part_n;
        tb ^= 1;
    }
#endif  // RG_DEVICE_BODY
}
