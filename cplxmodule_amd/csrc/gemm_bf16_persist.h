// Persistent form of the bf16 complex / real MFMA GEMM (gemm_bf16_impl.h): one workgroup per CU walks its
// output tiles, and the K-tile ring runs THROUGH the tile boundaries.
//
// Why (profiles/r02_gemm_ablation.md): with one workgroup per tile a CU pays, per output tile, the epilogue
// (~9 us), then -- because s_endpgm implies s_waitcnt 0 -- the drain of its stores, then the launch of the next
// workgroup and a cold prologue (all 256 CUs fetch their first two K tiles at once): ~23 us per round, four
// rounds per 8192 x 4096 x 4096 launch.  Here the first three K tiles of the NEXT output tile are requested
// while the current tile's last K tiles are computed, the epilogue stages its rows through the 16 KiB of LDS
// the ring leaves free, and the stores retire under the next tile's first MFMAs.
//
// s_waitcnt vmcnt counts loads AND stores of a wave in issue order, so a counted wait on an LDS-DMA piece
// issued after the epilogue's stores would also wait for those stores.  The schedule at a boundary is therefore:
//   last body of tile n : S5 issues ALL pieces of next-K2 (instead of half of them)
//   epilogue            : NST global stores per wave (exact count: full tiles only, no predicated stores)
//   body 0 of tile n+1  : S2 issues nothing;   S3 waits vmcnt(LOADS + NST)   (next-K1 is older than the stores)
//   body 1              : S3 waits vmcnt(LOADS + NST)                        (next-K2 is older than the stores)
//   body 2 ...          : the usual vmcnt(LOADS): first wait that includes the stores, ~2.5 K tiles later
// Ring slots stay compile-time constants: R = (K / 32) % 3 is a template parameter, K tile t of every output
// tile lives in slot (t + S0) % 3 with S0 = (3 - R) % 3, so the last three K tiles of a tile always sit in
// slots 0, 1, 2, the main loop is the same three bodies as in the one-tile kernel, and the next tile's K0..K2
// are simply sent to the slots (S0, S0+1, S0+2) % 3 as those free up (at most one of them a tile "late").
// (Two earlier forms -- ring positions relative to a per-tile rotation, and a run-time dispatch on R in front of
// the tail -- made the register allocator spill accumulator tiles at the joins.)
//
// The bias of the next tile (64 columns per wave) is fetched by ONE more LDS-DMA instruction per wave, issued
// before the stores (an ordinary load after them would make the compiler wait for vmcnt(0), i.e. for the stores).
// Preconditions (the launcher falls back to the one-tile-per-workgroup kernel otherwise): M % BM == 0,
// N % BN == 0, K / 32 >= 6, no split-K, no Gauss combine, no elementwise multiplier / accumulate operand,
// 16-byte aligned C rows and bias.
#pragma once

namespace cplxamd {

enum { PB_NORMAL = 0, PB_FIRST0 = 1, PB_FIRST1 = 2, PB_T3 = 3, PB_T2 = 4, PB_LAST = 5 };

template <typename TOUT, bool CPLX, bool CONJ, bool TA, bool TB, int R, bool FUSE = false>
__global__ __launch_bounds__((Cfg<CPLX>::NT)) void gemm_bf16_persist_kernel(GemmArgs g) {
  static_assert(!FUSE || sizeof(TOUT) == 2, "the fused LRT input gradient is a bf16-out kernel");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = Cfg<CPLX>;
  constexpr int NT = C::NT, BM = C::BM, BN = C::BN, IB = C::IB, JB = C::JB, L = C::LOADS;
  constexpr int H = (L + 1) / 2;
  constexpr int NPL = CPLX ? 2 : 1;
  constexpr int S0 = (3 - R) % 3;                 // slot of K tile 0 of every output tile
  constexpr int HEAD = S0 == 0 ? 3 : (S0 == 1 ? 2 : 4);   // K tiles in front of the slot-0-aligned main loop
  // global stores per wave and tile in the epilogue below (must be exact, see the header)
  constexpr int NST = sizeof(TOUT) == 2 ? NPL * IB * 4 : NPL * IB * JB * 4;
  static_assert(L + NST <= 63, "vmcnt is a 6-bit counter");

  const int tiles_m = g.M / BM, tiles_n = g.N / BN;
  const int ntiles = tiles_m * tiles_n;
  const int nwg = gridDim.x;
  // virtual block id -> tile origin (the XCD-contiguous grouped order of the one-tile kernel; nwg % 8 == 0
  // keeps "virtual block v runs on XCD v % 8" true for every tile of this workgroup)
  auto origin = [&](int v, int& m0, int& n0) __attribute__((always_inline)) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, idx = v >> 3;
    int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GM = g.group_m;
    const int per_group = GM * tiles_n;
    const int grp = lin / per_group, in_grp = lin - grp * per_group;
    const int first_m = grp * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    m0 = __builtin_amdgcn_readfirstlane((first_m + in_grp % gm) * BM);
    n0 = __builtin_amdgcn_readfirstlane((in_grp / gm) * BN);
  };

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid / C::WN) * (32 * IB), wn = (wid % C::WN) * (32 * JB);
  const int l31 = lane & 31, lk = lane >> 5;
  const int l15 = lane & 15, lg = (lane >> 4) & 1;

  const bf16_t* Ar = (const bf16_t*)g.a_r; const bf16_t* Ai = (const bf16_t*)g.a_i;
  const bf16_t* Br = (const bf16_t*)g.b_r; const bf16_t* Bi = (const bf16_t*)g.b_i;
  const int64_t lda = TA ? g.a_cs : g.a_rs, ldb = TB ? g.b_cs : g.b_rs;

  f32x16 acc_r[IB][JB], acc_i[CPLX ? IB : 1][JB];

  const uint32_t smem_off = lds_offset_of(smem);
  const uint32_t wave_lds = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 1024u;
  // per-lane byte offsets of the LDS-DMA pieces: full tiles only, so they do not depend on the tile
  uint32_t voa[C::PA], vob[C::PB];
#pragma unroll
  for (int j = 0; j < C::PA; ++j) voa[j] = piece_voff<BM, TA, NT>(lda, 0, BM, j);
#pragma unroll
  for (int j = 0; j < C::PB; ++j) vob[j] = piece_voff<BN, TB, NT>(ldb, 0, BN, j);

  const int nt = __builtin_amdgcn_readfirstlane(g.K / BK);

  // current / next tile origins; ring position of the current tile's K tile 0
  int v = blockIdx.x;
  int m0, n0, m0n, n0n;
  origin(v, m0, n0);
  bool has_next = v + nwg < ntiles;
  origin(has_next ? v + nwg : v, m0n, n0n);
  // ring slots (compile-time positions)
  const uint32_t soff[3] = {smem_off, smem_off + (uint32_t)C::STAGE_BYTES, smem_off + 2u * (uint32_t)C::STAGE_BYTES};
  const char* const sbase[3] = {smem, smem + C::STAGE_BYTES, smem + 2 * C::STAGE_BYTES};

  // LDS-DMA piece q into relative slot k.  `next` and the K index of a next-tile piece are compile-time facts of
  // the body that issues it (only the last three K tiles of an output tile reach into the next one), so no
  // run-time select or branch sits between the MFMA groups.  Without a next tile the "next" origin is this
  // tile's own: the surplus pieces re-read its K tiles 0..2 into slots nobody reads again.
  auto stage_piece = [&](uint32_t slot_off, bool next, int kk, int q) __attribute__((always_inline)) {
    const int mm = next ? m0n : m0, nn = next ? n0n : n0;
    const uint32_t s = slot_off + wave_lds;
    if (q < C::PA)
      lds_dma16_sv(piece_base<TA>(Ar, lda, mm, kk), voa[q], s + q * NT * 16);
    else if (q < C::PA + C::PB)
      lds_dma16_sv(piece_base<TB>(Br, ldb, nn, kk), vob[q - C::PA], s + C::A_BYTES + (q - C::PA) * NT * 16);
    else if (q < 2 * C::PA + C::PB)
      lds_dma16_sv(piece_base<TA>(Ai, lda, mm, kk), voa[q - C::PA - C::PB],
                   s + C::A_BYTES + C::B_BYTES + (q - C::PA - C::PB) * NT * 16);
    else
      lds_dma16_sv(piece_base<TB>(Bi, ldb, nn, kk), vob[q - 2 * C::PA - C::PB],
                   s + 2 * C::A_BYTES + C::B_BYTES + (q - 2 * C::PA - C::PB) * NT * 16);
  };

  auto a_frag = [&](const char* plane, int i, int ks) __attribute__((always_inline)) -> bf16x8 {
    if (TA) return frag_t<BM>(plane, wm + i * 32 + 16 * lg, ks * 16 + 8 * lk, l15);
    return frag_n(plane, wm + i * 32 + l31, ks * 2 + lk);
  };
  auto b_frag = [&](const char* plane, int j, int ks) __attribute__((always_inline)) -> bf16x8 {
    if (TB) return frag_t<BN>(plane, wn + j * 32 + 16 * lg, ks * 16 + 8 * lk, l15);
    return frag_n(plane, wn + j * 32 + l31, ks * 2 + lk);
  };

  bf16x8 ar[2][IB], br[2][JB], ai[2][IB], bi[2][JB];       // [ks][block]
  constexpr int NFRAG = CPLX ? 2 * IB + 4 : IB + JB;
  auto read_one = [&](const char* sA, int ks, int idx) __attribute__((always_inline)) {
    const char* sB = sA + C::A_BYTES;
    const char* sAi = sB + C::B_BYTES;
    const char* sBi = sAi + C::A_BYTES;
    if (CPLX) {
      if (idx == 0) br[ks][0] = b_frag(sB, 0, ks);
      else if (idx == 1) bi[ks][0] = b_frag(sBi, 0, ks);
      else if (idx == 2) ar[ks][0] = a_frag(sA, 0, ks);
      else if (idx == 3) ai[ks][0] = a_frag(sAi, 0, ks);
      else if (idx == 4) br[ks][1] = b_frag(sB, 1, ks);
      else if (idx == 5) bi[ks][1] = b_frag(sBi, 1, ks);
      else if ((idx & 1) == 0) ar[ks][(idx - 4) / 2] = a_frag(sA, (idx - 4) / 2, ks);
      else ai[ks][(idx - 5) / 2] = a_frag(sAi, (idx - 5) / 2, ks);
    } else {
      if (idx == 0) br[ks][0] = b_frag(sB, 0, ks);
      else if (idx == 1) ar[ks][0] = a_frag(sA, 0, ks);
      else if (idx == 2) br[ks][1] = b_frag(sB, 1, ks);
      else if (idx == 3) ar[ks][1] = a_frag(sA, 1, ks);
      else if (idx < 2 + JB) br[ks][idx - 2] = b_frag(sB, idx - 2, ks);
      else ar[ks][idx - JB] = a_frag(sA, idx - JB, ks);
    }
  };
  // 16 (real: IB*JB) MFMA groups on the fragments of sub-step ks; behind group n: a few fragment reads of
  // (relative slot rk, sub-step rks) and, for q in [q0, q1), LDS-DMA pieces of the K tile at element offset kk
  // (of this output tile, or of the next one) into relative slot dk
  auto mfma_half = [&](int ks, uint32_t dsoff, bool next, int kk, int q0, int q1, const char* rbase, int rks, bool do_read) __attribute__((always_inline)) {
    bf16x8 nai[IB];
    if (CPLX) {
#pragma unroll
      for (int i = 0; i < IB; ++i) nai[i] = neg_frag(CONJ ? ar[ks][i] : ai[ks][i]);
    }
    int q = q0;
    // complex: the two products into one accumulator are 8 MFMAs apart (blocks one after the other: +0.9 %)
    if constexpr (CPLX) {
      constexpr int NG = 2 * IB * JB;
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int i = 0; i < IB; ++i)
#pragma unroll
          for (int j = 0; j < JB; ++j) {
            if (ph == 0) {
              acc_r[i][j] = CPLXAMD_MFMA16(br[ks][j], ar[ks][i], acc_r[i][j]);
              acc_i[i][j] = CPLXAMD_MFMA16(br[ks][j], ai[ks][i], acc_i[i][j]);
            } else if (CONJ) {
              acc_r[i][j] = CPLXAMD_MFMA16(bi[ks][j], ai[ks][i], acc_r[i][j]);
              acc_i[i][j] = CPLXAMD_MFMA16(bi[ks][j], nai[i], acc_i[i][j]);
            } else {
              acc_r[i][j] = CPLXAMD_MFMA16(bi[ks][j], nai[i], acc_r[i][j]);
              acc_i[i][j] = CPLXAMD_MFMA16(bi[ks][j], ar[ks][i], acc_i[i][j]);
            }
            const int gidx = ph * IB * JB + i * JB + j;
            constexpr int PER = (NFRAG + NG - 1) / NG;
            const int g0 = gidx * PER;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < PER; ++r)
              if (do_read && g0 + r < NFRAG) read_one(rbase, rks, g0 + r);
            __builtin_amdgcn_sched_barrier(0);
            const int left = NG - gidx;
            int n_now = (q1 - q + left - 1) / left;
#pragma unroll
            for (int r = 0; r < 2; ++r)
              if (r < n_now && q < q1) {
                __builtin_amdgcn_sched_barrier(0);
                stage_piece(dsoff, next, kk, q);
                __builtin_amdgcn_sched_barrier(0);
                ++q;
              }
          }
      return;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        acc_r[i][j] = CPLXAMD_MFMA16(br[ks][j], ar[ks][i], acc_r[i][j]);
        if (CPLX) {
          acc_i[i][j] = CPLXAMD_MFMA16(br[ks][j], ai[ks][i], acc_i[i][j]);
          if (CONJ) {
            acc_r[i][j] = CPLXAMD_MFMA16(bi[ks][j], ai[ks][i], acc_r[i][j]);
            acc_i[i][j] = CPLXAMD_MFMA16(bi[ks][j], nai[i], acc_i[i][j]);
          } else {
            acc_r[i][j] = CPLXAMD_MFMA16(bi[ks][j], nai[i], acc_r[i][j]);
            acc_i[i][j] = CPLXAMD_MFMA16(bi[ks][j], ar[ks][i], acc_i[i][j]);
          }
        }
        {
          constexpr int PER = (NFRAG + IB * JB - 1) / (IB * JB);
          const int g0 = (i * JB + j) * PER;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < PER; ++r)
            if (do_read && g0 + r < NFRAG) read_one(rbase, rks, g0 + r);
          __builtin_amdgcn_sched_barrier(0);
        }
        // with more pieces than MFMA groups left, the surplus goes out behind the last group
        const int left = IB * JB - (i * JB + j);
        int n_now = (q1 - q + left - 1) / left;
#pragma unroll
        for (int r = 0; r < 2; ++r)
          if (r < n_now && q < q1) {
            __builtin_amdgcn_sched_barrier(0);
            stage_piece(dsoff, next, kk, q);
            __builtin_amdgcn_sched_barrier(0);
            ++q;
          }
      }
  };

  // one K tile t of the current output tile; KK = relative ring slot of this K tile, FL = what its LDS-DMA
  // pieces are (all compile time):
  //   flavour            S2: pieces [H, L) of           S5: pieces [0, H) of
  //   NORMAL / FIRST1    K tile t+2                     K tile t+3
  //   FIRST0             nothing (issued before)        K tile t+3
  //   T3 (t = nt-3)      K tile nt-1                    next tile's K0
  //   T2 (t = nt-2)      next tile's K0                 next tile's K1
  //   LAST (t = nt-1)    next tile's K1                 next tile's K2, ALL L pieces
  auto body = [&](auto KK, auto FL, int t) __attribute__((always_inline)) {
    constexpr int kc = decltype(KK)::value, fl = decltype(FL)::value;
    const uint32_t o0 = soff[kc], o2 = soff[(kc + 2) % 3];      // slots of K tile t and t + 2 (= t - 1)
    const char* p0 = sbase[kc];
    const char* p1 = sbase[(kc + 1) % 3];
    // next-tile K index that belongs in slot s: (s - S0) mod 3.  Tail: T3 sits in slot 0, T2 in 1, LAST in 2.
    constexpr int n0k = (0 - S0 + 3) % 3, n1k = (1 - S0 + 3) % 3, n2k = (2 - S0 + 3) % 3;
    if (fl == PB_FIRST0) mfma_half(0, o2, false, 0, 0, 0, p0, 1, true);
    else if (fl == PB_T2) mfma_half(0, o2, true, n0k * BK, H, L, p0, 1, true);          // -> slot 0
    else if (fl == PB_LAST) mfma_half(0, o2, true, n1k * BK, H, L, p0, 1, true);       // -> slot 1
    else mfma_half(0, o2, false, (t + 2) * BK, H, L, p0, 1, true);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // S3
    if (fl == PB_FIRST0 || fl == PB_FIRST1) wait_vmcnt<L + NST>(); else wait_vmcnt<L>();
    __builtin_amdgcn_s_barrier();
    if (fl == PB_T3) mfma_half(1, o0, true, n0k * BK, 0, H, p1, 0, true);              // -> slot 0
    else if (fl == PB_T2) mfma_half(1, o0, true, n1k * BK, 0, H, p1, 0, true);         // -> slot 1
    // (LAST: nothing may follow the stores; the next tile's first fragments are read AFTER the epilogue, so that
    //  they do not occupy 32 registers while the epilogue runs)
    else if (fl == PB_LAST) mfma_half(1, o0, true, n2k * BK, 0, L, p1, 0, false);      // -> slot 2, all L pieces
    else mfma_half(1, o0, false, (t + 3) * BK, 0, H, p1, 0, true);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;

  // ---- epilogue of the current tile: rows go through the 16 KiB above the ring (2 KiB per wave) so that a
  // store instruction writes eight whole 128-byte lines; NST unpredicated global stores per wave.
  // Lane-derived addresses of the per-tile code (epilogue, bias) are rebuilt from an OPAQUE copy of the thread
  // id each time: hoisted out of the tile loop they stayed live through the K loop and were spilled, and a
  // scratch reload behind the epilogue makes the compiler wait for vmcnt(0), i.e. for the stores.
  auto opaque_tid = [&]() __attribute__((always_inline)) -> int {
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    return tid;
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    const int tid = opaque_tid();
    const int lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lk = lane >> 5;
    const int wm = (wid / C::WN) * (32 * IB), wn = (wid % C::WN) * (32 * JB);
    char* reg = smem + 3 * C::STAGE_BYTES + wid * 2048;
    constexpr int PITCH = 144;                               // 128 B of payload + 16 B pad per staged row
    const int r8 = l31 >> 3, rr = l31 & 7;
    if constexpr (FUSE) {
      // C = acc + 2 X (*) ga (gemm.h).  The operands of two rounds (16 rows) -- 2 x (ga, x_r, x_i), 16 bytes per lane each,
      // the lane's own store position -- are requested together BEFORE those rounds' stores: vmcnt retires loads and
      // stores in issue order, so a load issued behind a store waits for the store's acknowledgement; batching makes
      // that happen three times per tile instead of once per round.
      const bf16_t* fxr = reinterpret_cast<const bf16_t*>(g.fx_r);
      const bf16_t* fxi = reinterpret_cast<const bf16_t*>(g.fx_i);
      const bf16_t* fga = reinterpret_cast<const bf16_t*>(g.fga);
      // batches of two rounds (16 rows): 6 loads = 24 registers in flight, 4 stores per batch
#pragma unroll
      for (int hb = 0; hb < IB * 2; ++hb) {
        const int i = hb >> 1, rbase = (hb & 1) * 2;
        uint4 lga[2], lxr[2], lxi[2];
        // address = wave-uniform base (scalar registers) + ONE 32-bit per-lane offset for all six loads
        const uint32_t loff = (uint32_t)(((lane >> 3) * (int)g.fld + (lane & 7) * 8) * 2);
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const int64_t ub = ((int64_t)(m0 + wm + i * 32 + (rbase + rd) * 8) * g.fld + n0 + wn) * 2;
          const uint32_t ulo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ub);
          const uint32_t uhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)ub >> 32));
          const int64_t u = (int64_t)(((uint64_t)uhi << 32) | ulo);
          lga[rd] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(fga) + u + loff);
          lxr[rd] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(fxr) + u + loff);
          if constexpr (CPLX) lxi[rd] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(fxi) + u + loff);
          else lxi[rd] = uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const int round = rbase + rd;
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            if (r8 == round) {
#pragma unroll
              for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  f4 x;
#pragma unroll
                  for (int e = 0; e < 4; ++e) x.v[e] = pl ? acc_i[CPLX ? i : 0][j][4 * q + e] : acc_r[i][j][4 * q + e];
                  st4(reinterpret_cast<bf16_t*>(reg + rr * PITCH + (j * 32 + 8 * q + 4 * lk) * 2), x);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint4 val = *reinterpret_cast<const uint4*>(reg + (lane >> 3) * PITCH + (lane & 7) * 16);
            const uint4 xv = pl ? lxi[rd] : lxr[rd];
            const uint32_t vw[4] = {val.x, val.y, val.z, val.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w};
            const uint32_t gw[4] = {lga[rd].x, lga[rd].y, lga[rd].z, lga[rd].w};
            uint32_t ow[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d0 = __uint_as_float(vw[e] << 16), d1 = __uint_as_float(vw[e] & 0xffff0000u);
              const float x0 = __uint_as_float(xw[e] << 16), x1 = __uint_as_float(xw[e] & 0xffff0000u);
              const float g0 = __uint_as_float(gw[e] << 16), g1 = __uint_as_float(gw[e] & 0xffff0000u);
              ow[e] = pack_bf16(fmaf(2.0f * x0, g0, d0), fmaf(2.0f * x1, g1, d1));
            }
            TOUT* out = reinterpret_cast<TOUT*>(pl ? g.c_i : g.c_r);
            const int row = m0 + wm + i * 32 + round * 8 + (lane >> 3), col = n0 + wn + (lane & 7) * 8;
            nt_store16(reinterpret_cast<bf16_t*>(out) + (int64_t)row * g.ldc + col, uint4{ow[0], ow[1], ow[2], ow[3]});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
        }
      }
      return;
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      TOUT* out = reinterpret_cast<TOUT*>(pl ? g.c_i : g.c_r);
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        if constexpr (sizeof(TOUT) == 2) {
#pragma unroll
          for (int round = 0; round < 4; ++round) {
            if (r8 == round) {
#pragma unroll
              for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  f4 x;
#pragma unroll
                  for (int e = 0; e < 4; ++e) x.v[e] = pl ? acc_i[CPLX ? i : 0][j][4 * q + e] : acc_r[i][j][4 * q + e];
                  st4(reinterpret_cast<bf16_t*>(reg + rr * PITCH + (j * 32 + 8 * q + 4 * lk) * 2), x);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint4 val = *reinterpret_cast<const uint4*>(reg + (lane >> 3) * PITCH + (lane & 7) * 16);
            const int row = m0 + wm + i * 32 + round * 8 + (lane >> 3), col = n0 + wn + (lane & 7) * 8;
            nt_store16(reinterpret_cast<bf16_t*>(out) + (int64_t)row * g.ldc + col, val);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
        } else {
#pragma unroll
          for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int round = 0; round < 4; ++round) {
              if (r8 == round) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  f4 x;
#pragma unroll
                  for (int e = 0; e < 4; ++e) x.v[e] = pl ? acc_i[CPLX ? i : 0][j][4 * q + e] : acc_r[i][j][4 * q + e];
                  st4(reinterpret_cast<float*>(reg + rr * PITCH + (8 * q + 4 * lk) * 4), x);
                }
              }
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              f4 x = ld4(reinterpret_cast<const float*>(reg + (lane >> 3) * PITCH + (lane & 7) * 16));
              const int row = m0 + wm + i * 32 + round * 8 + (lane >> 3), col = n0 + wn + j * 32 + (lane & 7) * 4;
              const int64_t o = (int64_t)row * g.ldc + col;
              float* of = reinterpret_cast<float*>(out);
              nt_store16(of + o, x);
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
      }
    }
  };

  // bias of the tile at column origin nn -> this wave's 512 bytes behind its staging rows (one LDS-DMA
  // instruction, lanes 0-15: real plane, 16-31: imaginary plane); without a bias a dummy read keeps the count
  const uint32_t bias_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_off + 3 * C::STAGE_BYTES + wave_lds * 2 + 1152));
  auto bias_dma = [&](int nn) __attribute__((always_inline)) {
    const int tid = opaque_tid();
    const int lane = tid & 63, wn = ((tid >> 6) % C::WN) * (32 * JB);
    if (lane < 16 * NPL) {
      const float* src = g.bias_r ? ((lane < 16 ? g.bias_r : g.bias_i) + nn + wn + 4 * (lane & 15))
                                  : reinterpret_cast<const float*>(Ar) + 4 * lane;
      lds_dma16_at(src, bias_lds);
    }
  };
  auto init_acc = [&]() __attribute__((always_inline)) {
    // acc = bias[n] (or 0): the bias rides in the accumulators
    const int tid = opaque_tid();
    const int wid = tid >> 6, lk = (tid & 63) >> 5;
    const char* breg = smem + 3 * C::STAGE_BYTES + wid * 2048 + 1152;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f4 b = {{0.f, 0.f, 0.f, 0.f}};
          if (g.bias_r) b = ld4(reinterpret_cast<const float*>(breg + pl * 256 + (j * 32 + 8 * q + 4 * lk) * 4));
#pragma unroll
          for (int i = 0; i < IB; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (pl) acc_i[CPLX ? i : 0][j][4 * q + e] = b.v[e];
              else acc_r[i][j][4 * q + e] = b.v[e];
            }
        }
  };

  // ---- prologue of the FIRST tile: K tiles 0, 1, 2 requested whole (the state every later tile starts from)
  bias_dma(n0);
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int q = 0; q < L; ++q) stage_piece(soff[(S0 + kt) % 3], false, kt * BK, q);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  init_acc();
  {
    const char* sA = sbase[S0];
#pragma unroll
    for (int idx = 0; idx < NFRAG; ++idx) read_one(sA, 0, idx);
  }

  for (;;) {
    // ---- K loop of the current output tile: HEAD tiles up to the slot-0 boundary, triples, the three tail tiles
    body(std::integral_constant<int, S0>{}, I1{}, 0);                    // FIRST0
    body(std::integral_constant<int, (S0 + 1) % 3>{}, I2{}, 1);          // FIRST1
    if constexpr (HEAD >= 3) body(std::integral_constant<int, (S0 + 2) % 3>{}, I0{}, 2);
    if constexpr (HEAD >= 4) body(std::integral_constant<int, (S0 + 3) % 3>{}, I0{}, 3);
    int t = HEAD;
    for (; t + 3 <= nt - 3; t += 3) {
      body(I0{}, I0{}, t);
      body(I1{}, I0{}, t + 1);
      body(I2{}, I0{}, t + 2);
    }
    body(I0{}, I3{}, t);                                                 // T3  (t == nt - 3)
    body(I1{}, I4{}, t + 1);                                             // T2
    body(I2{}, I5{}, t + 2);                                             // LAST
    bias_dma(has_next ? n0n : n0);     // (older than the stores below)
    epilogue();
    if (!has_next) break;
    wait_vmcnt<NST>();                 // everything issued BEFORE the stores has landed: next-K2 and the bias
    // ---- next output tile: its K tiles 0, 1 have landed or are in flight, F[0] holds (K0, ks 0) ---------
    v += nwg;
    m0 = m0n; n0 = n0n;
    has_next = v + nwg < ntiles;
    origin(has_next ? v + nwg : v, m0n, n0n);
    init_acc();
    {
      const char* sA = sbase[S0];
#pragma unroll
      for (int idx = 0; idx < NFRAG; ++idx) read_one(sA, 0, idx);
    }
  }
  wait_vmcnt<0>();   // surplus LDS-DMA pieces of the last tile's tail land before the LDS is released
}

}  // namespace cplxamd
