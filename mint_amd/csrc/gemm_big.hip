// Big-tile bf16 MFMA GEMM family for gfx950: ONE workgroup per CU, 8 waves, 32-deep K stages in a 4-deep
// LDS ring filled by 16-byte LDS-DMA (global_load_lds_dwordx4) and retired with COUNTED vmcnt waits, the
// two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) half a K step apart so that the LDS pipe
// (fragment reads + DMA issue of one group) and the MFMA pipe (the other group) of every SIMD run at the
// same time.  The structure was developed for the 288x256 NT kernel in round 1 (gemm.hip); here
// it is generalised over
//   * the tile shape: wave grid WGM x WGN (2x4 or 4x2), MR x NR MFMA tiles of 16x16 per wave
//       288x256 (2,9,4,4)   256x256 (2,8,4,4)   N = 3072 / 2400 outputs (FFN1, GELU' dgrad, QKV)
//       256x160 (4,4,2,5)                       N = 800 outputs: 800 = 5 x 160 exactly, and one wave's 80
//                                               columns are exactly one attention head (dh = 80)
//       160x256 (2,5,4,4)                       weight gradients with an 800-row side
//   * the operand layout:
//       NT  A[m][k], B[n][k]  (contraction contiguous)  forward / dgrad   - ds_read_b128 fragments
//       TN  A[k][m], B[k][n]  (contraction = tokens)    wgrad             - ds_read_b64_tr_b16 fragments
//     so the weight gradients get the same pipeline as the forward GEMMs and need no split-K: a layer's
//     four wgrad GEMMs are ONE grouped launch of 190 whole-K tiles (fp32 read-modify-write of the grad
//     arena, no slabs, no reduce pass).
// MFMA roles are swapped (N-side fragment as the A operand) so that a lane owns 4 consecutive output
// columns; a problem flagged `trans_out` (dW stored as [n][m]) uses the un-swapped roles instead, which
// leaves 4 consecutive m per lane = 16 contiguous bytes of the transposed output.
#include "gemm.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

DEVINL void glds16(const bf16_t* src, unsigned char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst, 16, 0, 0);
}
template <int N>
DEVINL void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `stages_after` later stages (LPS loads each) are still in flight
template <int LPS>
DEVINL void wait_stages(int stages_after) {
  switch (stages_after) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<LPS>(); break;
    case 2: wait_vmcnt<2 * LPS>(); break;
    default: wait_vmcnt<3 * LPS>(); break;
  }
}
DEVINL unsigned lds_addr(const unsigned char* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
template <int OFF>
DEVINL bf16x4 tr_read(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  bf16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

// MF (round 6): MFMA shape of the NT kernels - 16 = v_mfma_f32_16x16x32_bf16 (MR x NR tiles of 16 x 16 per wave), 32 =
// v_mfma_f32_32x32x16_bf16 (MR x NR tiles of 32 x 32; 64-deep ring slots only): 32 x 32 x 16 issues in 32 cycles for twice the
// FLOPs of a 17-19-cycle 16 x 16 x 32 (tools/mfma_rate_probe.hip: 2 068 vs 1 889 TFLOP/s chip-wide).  Accumulator layout of the
// 32 x 32 form (swapped roles: rows of D = output columns n): lane = output row m = lane & 31, register 4 q + r = output
// column 8 q + 4 (lane >> 5) + r - four QUADS of 4 consecutive columns per tile where the 16 x 16 form has one.
template <int WGM_, int MR_, int WGN_, int NR_, int NSTAGE_ = 4, int WGS_PER_CU_ = 1, int KS_ = 32, int MF_ = 16>
struct BigCfg {
  static constexpr int WGM = WGM_, MR = MR_, WGN = WGN_, NR = NR_, MF = MF_;
  static_assert(MF == 16 || (MF == 32 && KS_ == 64), "MFMA shape");
  static constexpr int BM = WGM * MR * MF, BN = WGN * NR * MF;
  static constexpr int WROWS = MR * MF, WCOLS = NR * MF;  // wave tile
  static constexpr int NQ = NR * ((MF == 32) ? 4 : 1);  // quads of 4 consecutive columns per accumulator row: NR (16) / 4 NR (32)
  static constexpr int QPT = (MF == 32) ? 4 : 1;                   // quads per tile
  using AccT = std::conditional_t<MF == 32, f32x16, f32x4>;       // accumulator of one MFMA tile
  // first column (inside the wave tile) of quad jq for this lane; the lane's row inside a row tile is lane & (MF - 1)
  __device__ __forceinline__ static int qcol(int jq, int lane) {
    if constexpr (MF == 32) return (jq >> 2) * 32 + (jq & 3) * 8 + (lane >> 5) * 4;
    else return jq * 16 + (lane >> 4) * 4;
  }
  static constexpr int NSTAGE = NSTAGE_, DIST = NSTAGE - 1;
  static constexpr int WGS_PER_CU = WGS_PER_CU_;  // co-resident workgroups the register / LDS budget is sized for
  // KS = depth of a ring slot along K.  32 (rounds 1-2): a 1 KiB DMA piece is 16 rows x 64 B - HALF a 128-byte line per
  // row, so every operand line is requested twice, one K step apart (round-3 PMC: 3.61 M L2 read requests for the
  // FFN1-shaped GEMM against 1.63 M lines of operand bytes; the 64-deep 128x128 kernel: 3.53 M for 3.46 M lines).
  // 64 (big_mainloop64, NT only): a piece is 8 rows x 128 B = whole lines; a slot feeds two 32-deep multiply steps.
  static constexpr int KS = KS_;
  static_assert(KS == 32 || KS == 64, "slot depth");
  static constexpr int PROWS = (KS == 64) ? 8 : 16;  // rows per DMA piece
  static constexpr int A_PIECES = BM / PROWS, B_PIECES = BN / PROWS;  // 1 KiB DMA pieces per stage
  static constexpr int A_BYTES = A_PIECES * 1024;
  static constexpr int NPIECE = A_PIECES + B_PIECES;
  static constexpr int STAGE_BYTES = NPIECE * 1024;
  static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
  static constexpr int NW = WGM * WGN;  // 8 waves: two groups half a K step apart
  static constexpr int LPS_LO = NPIECE / NW, EXTRA = NPIECE % NW;  // waves < EXTRA issue one more piece
  static_assert(NW == 8, "8 waves");
  static_assert((NSTAGE >= 3 || KS_ == 64) && NSTAGE >= 2 && NSTAGE <= 6, "ring depth");
  static_assert(LDS_BYTES * WGS_PER_CU <= 160 * 1024, "LDS");
};

// ---- NT stage image: [rows][32 k] = 64-byte rows; logical 16-byte chunk c of row r sits at position
// c ^ G[(r>>2)&3], G = {0,2,3,1}: every ds_read_b128 service group lands on 16 distinct 16-byte bank slots.
DEVINL int ring_g(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }
DEVINL bf16x8 ring_frag(const unsigned char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((chunk ^ ring_g(row)) << 4));
}

// ---- TN stage image of one operand ([32 k][W columns], W = 128*n128 + 32*tail):
//   n128 sub-images [32 k][128 cols] with 256-byte rows: logical 32-byte unit u of row k at unit position
//     u ^ f(k), f(k) = (k&3) | ((k>>3)&1)<<2  (the layout of the 128x128 TN kernel of gemm.hip): the 8 row
//     segments one ds_read_b64_tr_b16 half-wave touches fall in 8 distinct 32-byte bank slots;
//   then, if W % 128 == 32, one sub-image [32 k][32 cols] with 64-byte rows: unit u (0/1) of row k at
//     position u ^ ((k>>3)&1)  (rows k..k+3 cover slots {0,2,4,6} + u, rows k+8.. the other parity).
// A 16-column MFMA tile is one unit.  Fragment for unit U: lane group g = lane>>4, s = lane&15 supplies the
// address of 4 contiguous columns (s&3)*4.. of row k = g*8 + hh*4 + (s>>2); the hardware returns to lane c
// the 4 k-values of column c (tests/test_gpu_ops.py probe) -> two reads (hh = 0, 1) give the 8 k-slots.
DEVINL int tn_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <int W>
struct TnImg {
  static constexpr int N128 = W / 128, TAIL = (W % 128) / 32;
  static_assert(W % 128 == 0 || W % 128 == 32, "TN operand width");
  static constexpr int PIECES = W / 16;
  // Units (16-column MFMA tiles) of the waves along this operand: wave w of WG owns R units, the first
  // MAIN of them from the 128-wide sub-images and the rest from the 32-wide tail, so that the hh row
  // increment of fragment i (256-byte vs 64-byte rows) is a compile-time constant of i.
  template <int WG, int R>
  DEVINL static int unit_of(int w, int i) {
    constexpr int MAIN = 8 * N128 / WG, TAILP = R - MAIN;
    static_assert((8 * N128) % WG == 0 && TAILP * WG == 2 * TAIL, "unit split");
    return i < MAIN ? w * MAIN + i : 8 * N128 + w * TAILP + (i - MAIN);
  }
  template <int WG, int R>
  static constexpr int dhh_of(int i) { return i < 8 * N128 / WG ? 4 * 256 : 4 * 64; }
  // source column (relative to the tile) and k row that lane `lane` of DMA piece `p` must fetch
  DEVINL static void piece_src(int p, int lane, int& k, int& col) {
    if (p < N128 * 8) {
      const int sub = p >> 3, pp = p & 7;
      k = pp * 4 + (lane >> 4);
      const int lp = lane & 15;
      col = sub * 128 + ((((lp >> 1) ^ tn_f(k)) << 1) | (lp & 1)) * 8;
    } else {
      const int tp = p - N128 * 8;
      k = tp * 16 + (lane >> 2);
      const int cp = lane & 3;
      col = N128 * 128 + ((((cp >> 1) ^ ((k >> 3) & 1)) << 1) | (cp & 1)) * 8;
    }
  }
  // byte offset (inside the operand image) of the hh = 0 read of unit U, and the hh = 1 increment
  DEVINL static void frag_off(int U, int lane, unsigned& off0, unsigned& dhh) {
    const int g = lane >> 4, s = lane & 15;
    if (U < N128 * 8) {
      const int sub = U >> 3, u = U & 7;
      const int f = ((s >> 2) & 3) | ((g & 1) << 2);
      off0 = (unsigned)(sub * 8192 + (g * 8 + (s >> 2)) * 256 + ((u ^ f) << 5) + (s & 3) * 8);
      dhh = 4 * 256;
    } else {
      const int u = U - N128 * 8;
      off0 = (unsigned)(N128 * 8192 + (g * 8 + (s >> 2)) * 64 + ((u ^ (g & 1)) << 5) + (s & 3) * 8);
      dhh = 4 * 64;
    }
  }
};

template <class C, int SO, int PAIR, int J>
DEVINL void tn_reads_b(const unsigned (&tb)[C::NR][(C::NSTAGE + 1) / 2], bf16x4 (&blo)[C::NR], bf16x4 (&bhi)[C::NR]) {
  if constexpr (J < C::NR) {
    constexpr int DH = TnImg<C::BN>::template dhh_of<C::WGN, C::NR>(J);
    blo[J] = tr_read<SO>(tb[J][PAIR]);
    bhi[J] = tr_read<SO + DH>(tb[J][PAIR]);
    tn_reads_b<C, SO, PAIR, J + 1>(tb, blo, bhi);
  }
}
template <class C, int SO, int PAIR, int I>
DEVINL void tn_reads_a(const unsigned (&ta)[C::MR][(C::NSTAGE + 1) / 2], bf16x4 (&alo)[C::MR], bf16x4 (&ahi)[C::MR]) {
  if constexpr (I < C::MR) {
    constexpr int DH = TnImg<C::BM>::template dhh_of<C::WGM, C::MR>(I);
    alo[I] = tr_read<SO>(ta[I][PAIR]);
    ahi[I] = tr_read<SO + DH>(ta[I][PAIR]);
    tn_reads_a<C, SO, PAIR, I + 1>(ta, alo, ahi);
  }
}
// transpose read number R (0 .. 2*(NR+MR)-1) of a TN stage: B units first, then A units; even R = hh 0, odd = hh 1
template <class C, int SO, int PAIR, int R>
DEVINL void tn_read_one(const unsigned (&ta)[C::MR][(C::NSTAGE + 1) / 2], const unsigned (&tb)[C::NR][(C::NSTAGE + 1) / 2], bf16x4 (&alo)[C::MR],
                        bf16x4 (&ahi)[C::MR], bf16x4 (&blo)[C::NR], bf16x4 (&bhi)[C::NR]) {
  if constexpr (R < 2 * C::NR) {
    constexpr int J = R / 2;
    constexpr int DH = TnImg<C::BN>::template dhh_of<C::WGN, C::NR>(J);
    if constexpr (R % 2 == 0) blo[J] = tr_read<SO>(tb[J][PAIR]);
    else bhi[J] = tr_read<SO + DH>(tb[J][PAIR]);
  } else if constexpr (R < 2 * (C::NR + C::MR)) {
    constexpr int I = (R - 2 * C::NR) / 2;
    constexpr int DH = TnImg<C::BM>::template dhh_of<C::WGM, C::MR>(I);
    if constexpr (R % 2 == 0) alo[I] = tr_read<SO>(ta[I][PAIR]);
    else ahi[I] = tr_read<SO + DH>(ta[I][PAIR]);
  }
}
DEVINL void mfma16_asm(f32x4& c, bf16x8 a, bf16x8 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// MFMA number X of a step (row-major over the MR x NR tile) followed by transpose read number X of the next stage
template <class C, bool SWAP, int SO, int PAIR, int X, int ABL = 0>
DEVINL void tn_mfma_read_chain(f32x4 (&acc)[C::MR][C::NR], const bf16x8 (&af)[C::MR], const bf16x8 (&bfr)[C::NR],
                               const unsigned (&ta)[C::MR][(C::NSTAGE + 1) / 2], const unsigned (&tb)[C::NR][(C::NSTAGE + 1) / 2], bf16x4 (&alo)[C::MR],
                               bf16x4 (&ahi)[C::MR], bf16x4 (&blo)[C::NR], bf16x4 (&bhi)[C::NR]) {
  if constexpr (X < C::MR * C::NR) {
    constexpr int I = X / C::NR, J = X % C::NR;
    if constexpr (ABL != 2) {  // ABL 2: no MFMAs
      if constexpr (SWAP) mfma16_asm(acc[I][J], bfr[J], af[I]);
      else mfma16_asm(acc[I][J], af[I], bfr[J]);
    }
    if constexpr (ABL != 3) tn_read_one<C, SO, PAIR, X>(ta, tb, alo, ahi, blo, bhi);  // ABL 3: no fragment reads
    tn_mfma_read_chain<C, SWAP, SO, PAIR, X + 1, ABL>(acc, af, bfr, ta, tb, alo, ahi, blo, bhi);
  }
}

// The 2 * (MR + NR) transpose reads of one TN stage: B units first (the multiply needs them for every MFMA).
template <class C, int SO, int PAIR>
DEVINL void tn_reads(const unsigned (&ta)[C::MR][(C::NSTAGE + 1) / 2], const unsigned (&tb)[C::NR][(C::NSTAGE + 1) / 2], bf16x4 (&alo)[C::MR],
                     bf16x4 (&ahi)[C::MR], bf16x4 (&blo)[C::NR], bf16x4 (&bhi)[C::NR]) {
  tn_reads_b<C, SO, PAIR, 0>(tb, blo, bhi);
  tn_reads_a<C, SO, PAIR, 0>(ta, alo, ahi);
}

// -------------------------------------------------------------------------------------------------
// Shared main loop.  This wave's DMA pieces are described by a wave-uniform source pointer `sptr[i]` (stage 0;
// advanced by `sadv[i]` bytes per stage with SALU adds), a per-lane byte offset `voff[i]` and the byte offset
// `dst[i]` inside a stage: the loads are issued in the scalar-base + 32-bit-vector-offset form, no per-lane
// pointer arithmetic in the loop.  The K loop is unrolled over the NSTAGE = 4 ring slots so that every LDS
// address is a loop-invariant VGPR plus an IMMEDIATE (ds offset field): the read phase of a wave is nothing
// but its DMA issues and fragment reads.  Stages past the end of K are still issued (from the last valid
// stage, into slots nobody reads any more), which makes every wait a constant vmcnt(2 stages).
// On exit acc holds the wave's sub-tile; SWAP: swapped MFMA roles (lane owns 4 consecutive n of one m).
// Round-2 PMC on the first version of this loop (per-lane 64-bit pointers, runtime ring slot): 44 VALU + 34
// SALU beside 20 MFMAs per wave and K step, read phase ~2x the multiply phase, 37 % MFMA busy (TN 160x256).
// -------------------------------------------------------------------------------------------------
template <int S>
struct SlotC { static constexpr int value = S; };
// (Rounds 2-5 carried build knobs here - DMA issues ahead of the fragment reads, static / no s_setprio, DMA pieces issued from
// inside the multiply phase, write-through / non-temporal epilogue stores: all measured neutral or worse, DESIGN section 6 -
// removed in round 6; what is left is the shipped schedule.)

// CS (TN only): the waves with `do_cs` also accumulate cs[x] += X_x * ones per K step - the column sums over k of the operand
// on the MFMA-A side (SWAP: the B units bfr[j], else the A units af[i]); row r of cs[x] in every lane column is
// sum_k operand[k][unit x, column (lane >> 4) * 4 + r].  The multiply phase of this loop has slack under the other group's
// read phase (TN: transpose reads, ~2x the multiply phase), which is where the extra MFMAs go.
template <class C, bool TN, bool SWAP, bool CS = false>
DEVINL void big_mainloop(unsigned char* smem, const char* (&sptr)[C::LPS_LO + 1], const unsigned (&sadv)[C::LPS_LO + 1],
                         const unsigned (&voff)[C::LPS_LO + 1], const int (&dst)[C::LPS_LO + 1], int nk, int wave,
                         int wm, int wn, int lane, f32x4 (&acc)[C::MR][C::NR], bool do_cs = false, f32x4* cs = nullptr) {
  constexpr int MR = C::MR, NR = C::NR, NST = C::NSTAGE, DIST = C::DIST, STAGE = C::STAGE_BYTES;
  constexpr int LPS_LO = C::LPS_LO, EXTRA = C::EXTRA;
  const int grp = wave >> 2;  // stagger group: waves 0-3 lead, waves 4-7 run one phase behind
  const bool extra = EXTRA && wave < EXTRA;

  auto stage = [&](auto slot_c, bool more) {
    constexpr int SLOT = decltype(slot_c)::value;
    unsigned char* base = smem + SLOT * STAGE;
#pragma unroll
    for (int i = 0; i < LPS_LO; ++i) glds16(reinterpret_cast<const bf16_t*>(sptr[i] + voff[i]), base + dst[i]);
    if (extra) glds16(reinterpret_cast<const bf16_t*>(sptr[LPS_LO] + voff[LPS_LO]), base + dst[LPS_LO]);
#pragma unroll
    for (int i = 0; i < LPS_LO + 1; ++i) sptr[i] += more ? sadv[i] : 0u;
  };
  auto wait_ahead = [&]() {  // at most DIST - 1 later stages of this wave stay in flight
    if (extra) wait_vmcnt<(DIST - 1) * (LPS_LO + 1)>();
    else wait_vmcnt<(DIST - 1) * LPS_LO>();
  };
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // loop-invariant fragment addresses
  unsigned a_nt[NST], b_nt[NST];          // NT: per ring slot, + i * 1024 immediates
  unsigned ta[MR][(NST + 1) / 2], tb[NR][(NST + 1) / 2];  // TN: per unit and slot pair {0,1} / {2,3}, + immediates
  if constexpr (!TN) {
    const int r = lane & 15, chunk = lane >> 4;
    const unsigned lanepart = (unsigned)(r * 64 + ((chunk ^ ring_g(r)) << 4));
#pragma unroll
    for (int sl = 0; sl < NST; ++sl) {
      a_nt[sl] = (unsigned)(sl * STAGE + wm * MR * 1024) + lanepart;
      b_nt[sl] = (unsigned)(sl * STAGE + C::A_BYTES + wn * NR * 1024) + lanepart;
    }
  } else {
    static_assert(!TN || NST == 4, "TN slot-pair addressing assumes 4 ring slots");
    const unsigned l0 = lds_addr(smem);
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      unsigned off, dh;
      TnImg<C::BM>::frag_off(TnImg<C::BM>::template unit_of<C::WGM, MR>(wm, i), lane, off, dh);
      ta[i][0] = l0 + off;
      ta[i][1] = l0 + off + 2 * STAGE;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      unsigned off, dh;
      TnImg<C::BN>::frag_off(TnImg<C::BN>::template unit_of<C::WGN, NR>(wn, j), lane, off, dh);
      tb[j][0] = l0 + C::A_BYTES + off;
      tb[j][1] = l0 + C::A_BYTES + off + 2 * STAGE;
    }
  }

  stage(SlotC<0>{}, 1 < nk);
  stage(SlotC<1>{}, 2 < nk);
  if constexpr (DIST == 3) stage(SlotC<2>{}, 3 < nk);
  wait_ahead();
  __builtin_amdgcn_s_barrier();  // stage 0 landed

  //   phase 2k   : group 0 stages k+DIST and reads k   | group 1 multiplies k-1
  //   phase 2k+1 : group 0 multiplies k                | group 1 stages k+DIST and reads k
  // Every wave retires its own pieces of stage k+1 (counted vmcnt) before the barrier that ends phase 2k+1
  // and its fragment reads (lgkmcnt) before the barrier that ends its read phase, so a ring slot is only
  // re-armed after both groups are done with it (derivation: gemm.hip gemm_nt_big_kernel).
  bf16x8 af[MR], bfr[NR];
  auto body = [&](auto slot_c, int kt) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int NEXT = (SLOT + DIST) % NST;  // ring slot of stage kt + DIST (= the slot stage kt - 1 used)
    // fragment reads first, the DMA issues of stage kt+DIST (another ring slot) behind them: the ~60-100 cycles
    // each LDS-DMA instruction takes to issue then cover the LDS latency of the reads instead of preceding it
    if constexpr (!TN) {
#pragma unroll
      for (int j = 0; j < NR; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(smem + b_nt[SLOT] + j * 1024);
#pragma unroll
      for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const bf16x8*>(smem + a_nt[SLOT] + i * 1024);
      __builtin_amdgcn_sched_barrier(0);
      stage(SlotC<NEXT>{}, kt + DIST + 1 < nk);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      constexpr int SO = (SLOT & 1) * STAGE, PAIR = SLOT / 2;
      bf16x4 blo[NR], bhi[NR], alo[MR], ahi[MR];
      tn_reads<C, SO, PAIR>(ta, tb, alo, ahi, blo, bhi);
      stage(SlotC<NEXT>{}, kt + DIST + 1 < nk);
      // the asm reads are invisible to hipcc's counters: retire them by hand and pin the order
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NR; ++j) bfr[j] = __builtin_shufflevector(blo[j], bhi[j], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int i = 0; i < MR; ++i) af[i] = __builtin_shufflevector(alo[i], ahi[i], 0, 1, 2, 3, 4, 5, 6, 7);
    }
    if (grp == 1) wait_ahead();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) acc[i][j] = SWAP ? mfma16(bfr[j], af[i], acc[i][j]) : mfma16(af[i], bfr[j], acc[i][j]);
    if constexpr (CS) {
      if (do_cs) {  // wave-uniform
        const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
#pragma unroll
        for (int x = 0; x < (SWAP ? NR : MR); ++x) cs[x] = mfma16(SWAP ? bfr[x] : af[x], ones, cs[x]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (grp == 0) wait_ahead();
    __builtin_amdgcn_s_barrier();
  };
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one phase behind group 0
  for (int kt = 0; kt < nk; kt += NST) {
    body(SlotC<0>{}, kt);
    if (kt + 1 < nk) body(SlotC<1>{}, kt + 1);
    if (kt + 2 < nk) body(SlotC<2>{}, kt + 2);
    if constexpr (NST == 4)
      if (kt + 3 < nk) body(SlotC<3>{}, kt + 3);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();  // the run-ahead stages past the end of K: landed before the ring is reused for staging
  __builtin_amdgcn_s_barrier();
}

// -------------------------------------------------------------------------------------------------
// TN main loop with FRAGMENT PREFETCH (round 2b).  PMC on the staggered loop above, TN 160x256: the transpose
// reads deliver 8 bytes per lane, so a wave needs 18 of them per K step and issues one every ~15 cycles
// (SQ_WAIT_INST_LDS 15 % of wave cycles) - its read phase (~600 cycles) is twice the multiply phase of the
// other group (320), MFMA busy 44 %.  Here a wave reads the fragments of stage k+1 into a second register set
// WHILE it multiplies stage k: the reads are off its critical path, the two waves of a SIMD drift apart by
// themselves (one issues LDS reads while the other issues MFMAs) and one barrier per K step is left:
//   top of step k : wait for this wave's DMA pieces of stage k+1 (one later stage may stay in flight), barrier
//   body          : DMA issues of stage k+3 (ring slot of stage k-1, whose reads ended before the barrier),
//                   transpose reads of stage k+1 -> F[(k+1)&1], 20 MFMAs on F[k&1], lgkmcnt(0)
// -------------------------------------------------------------------------------------------------
template <class C, bool SWAP, bool FINE, int ABL = 0>
DEVINL void tn_mainloop_pf(unsigned char* smem, const char* (&sptr)[C::LPS_LO + 1],
                           const unsigned (&sadv)[C::LPS_LO + 1], const unsigned (&voff)[C::LPS_LO + 1],
                           const int (&dst)[C::LPS_LO + 1], int nk, int wave, int wm, int wn, int lane,
                           f32x4 (&acc)[C::MR][C::NR]) {
  constexpr int MR = C::MR, NR = C::NR, STAGE = C::STAGE_BYTES, NS = C::NSTAGE, NPAIR = (NS + 1) / 2;
  constexpr int LPS_LO = C::LPS_LO, EXTRA = C::EXTRA;
  // ring of NS slots: stage k+NS-1 is issued at step k (into the slot stage k-1 used); the wait at the top of step
  // k leaves the NS-3 youngest stages in flight (NS = 4: one, the round-2b loop; NS = 6: three - the deeper ring
  // rides out the longer L2 / fabric latency the kernel sees when it shares the chip with the dgrad chain)
  static_assert(NS >= 4 && NS <= 6, "ring depth of the prefetch loop");
  const bool extra = EXTRA && wave < EXTRA;
  auto stage = [&](auto slot_c, bool more) {
    constexpr int SLOT = decltype(slot_c)::value;
    unsigned char* base = smem + SLOT * STAGE;
#pragma unroll
    for (int i = 0; i < LPS_LO; ++i) glds16(reinterpret_cast<const bf16_t*>(sptr[i] + voff[i]), base + dst[i]);
    if (extra) glds16(reinterpret_cast<const bf16_t*>(sptr[LPS_LO] + voff[LPS_LO]), base + dst[LPS_LO]);
#pragma unroll
    for (int i = 0; i < LPS_LO + 1; ++i) sptr[i] += more ? sadv[i] : 0u;
  };
  auto wait_top = [&]() {  // at most NS - 3 later stages of this wave stay in flight
    if (extra) wait_vmcnt<(NS - 3) * (LPS_LO + 1)>();
    else wait_vmcnt<(NS - 3) * LPS_LO>();
  };
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned ta[MR][NPAIR], tb[NR][NPAIR];  // per slot pair {2p, 2p+1}: + (slot & 1) * STAGE as an immediate
  {
    const unsigned l0 = lds_addr(smem);
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      unsigned off, dh;
      TnImg<C::BM>::frag_off(TnImg<C::BM>::template unit_of<C::WGM, MR>(wm, i), lane, off, dh);
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr) ta[i][pr] = l0 + off + 2 * pr * STAGE;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      unsigned off, dh;
      TnImg<C::BN>::frag_off(TnImg<C::BN>::template unit_of<C::WGN, NR>(wn, j), lane, off, dh);
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr) tb[j][pr] = l0 + C::A_BYTES + off + 2 * pr * STAGE;
    }
  }
  // two fragment register sets
  bf16x4 alo0[MR], ahi0[MR], blo0[NR], bhi0[NR], alo1[MR], ahi1[MR], blo1[NR], bhi1[NR];

  stage(SlotC<0>{}, 1 < nk);
  stage(SlotC<1>{}, 2 < nk);
  stage(SlotC<2>{}, 3 < nk);
  if constexpr (NS >= 5) stage(SlotC<3>{}, 4 < nk);
  if constexpr (NS >= 6) stage(SlotC<4>{}, 5 < nk);
  if (extra) wait_vmcnt<(NS - 2) * (LPS_LO + 1)>();
  else wait_vmcnt<(NS - 2) * LPS_LO>();
  __builtin_amdgcn_s_barrier();  // stage 0 landed
  tn_reads<C, 0, 0>(ta, tb, alo0, ahi0, blo0, bhi0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // `par` = parity of the step (which fragment register set holds stage kt); the ring slot is compile time
  auto body = [&](auto slot_c, auto par_c, int kt) {
    constexpr int SLOT = decltype(slot_c)::value;     // ring slot of stage kt
    constexpr int PAR = decltype(par_c)::value;
    constexpr int NSL = (SLOT + 1) % NS;              // ring slot of stage kt + 1
    constexpr int SO = (NSL & 1) * STAGE, PAIR = NSL / 2;
    if constexpr (ABL != 1) wait_top();
    __builtin_amdgcn_s_barrier();
    if constexpr (ABL != 1) stage(SlotC<(SLOT + NS - 1) % NS>{}, kt + NS < nk);  // ABL 1: no DMA in the loop
    if constexpr (!FINE) {
      if constexpr (PAR == 0) tn_reads<C, SO, PAIR>(ta, tb, alo1, ahi1, blo1, bhi1);
      else tn_reads<C, SO, PAIR>(ta, tb, alo0, ahi0, blo0, bhi0);
    }
    bf16x8 af[MR], bfr[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j)
      bfr[j] = PAR ? __builtin_shufflevector(blo1[j], bhi1[j], 0, 1, 2, 3, 4, 5, 6, 7)
                   : __builtin_shufflevector(blo0[j], bhi0[j], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int i = 0; i < MR; ++i)
      af[i] = PAR ? __builtin_shufflevector(alo1[i], ahi1[i], 0, 1, 2, 3, 4, 5, 6, 7)
                  : __builtin_shufflevector(alo0[i], ahi0[i], 0, 1, 2, 3, 4, 5, 6, 7);
    if constexpr (FINE) {
      // one transpose read of stage k+1 behind every MFMA of stage k (both as asm: program order is issue order)
      static_assert(2 * (MR + NR) <= MR * NR, "a read slot per MFMA");
      if constexpr (PAR == 0)
        tn_mfma_read_chain<C, SWAP, SO, PAIR, 0, ABL>(acc, af, bfr, ta, tb, alo1, ahi1, blo1, bhi1);
      else
        tn_mfma_read_chain<C, SWAP, SO, PAIR, 0, ABL>(acc, af, bfr, ta, tb, alo0, ahi0, blo0, bhi0);
    } else {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
          acc[i][j] = SWAP ? mfma16(bfr[j], af[i], acc[i][j]) : mfma16(af[i], bfr[j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    }
    // the prefetched fragments must have landed before the next step multiplies them (asm reads are
    // invisible to hipcc's counters) - and before this wave passes the next barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // unrolled over lcm(NS, 2) steps so that both the ring slot and the fragment-set parity are compile time
  constexpr int UNR = (NS % 2 == 0) ? NS : 2 * NS;
  for (int kt = 0; kt < nk; kt += UNR) {
    body(SlotC<0>{}, SlotC<0>{}, kt);
    if (kt + 1 < nk) body(SlotC<1 % NS>{}, SlotC<1>{}, kt + 1);
    if (kt + 2 < nk) body(SlotC<2 % NS>{}, SlotC<0>{}, kt + 2);
    if (kt + 3 < nk) body(SlotC<3 % NS>{}, SlotC<1>{}, kt + 3);
    if constexpr (UNR > 4) {
      if (kt + 4 < nk) body(SlotC<4 % NS>{}, SlotC<0>{}, kt + 4);
      if (kt + 5 < nk) body(SlotC<5 % NS>{}, SlotC<1>{}, kt + 5);
    }
    if constexpr (UNR > 6) {
      if (kt + 6 < nk) body(SlotC<6 % NS>{}, SlotC<0>{}, kt + 6);
      if (kt + 7 < nk) body(SlotC<7 % NS>{}, SlotC<1>{}, kt + 7);
      if (kt + 8 < nk) body(SlotC<8 % NS>{}, SlotC<0>{}, kt + 8);
      if (kt + 9 < nk) body(SlotC<9 % NS>{}, SlotC<1>{}, kt + 9);
    }
  }
  wait_vmcnt<0>();  // run-ahead stages past the end of K
  // asm MFMAs: the accumulators are read by the compiler's epilogue code - cover the XDL write-back latency
  if constexpr (FINE) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// -------------------------------------------------------------------------------------------------
// NT main loop on 64-DEEP ring slots (round 3).  Same two staggered wave groups and the same 32-deep multiply steps as
// big_mainloop, but a ring slot now holds 64 k of every row as ONE 128-byte line (image: 16-byte chunk c of row r at
// position c ^ (r & 7), the layout of the 128x128 kernel - conflict-free ds_read_b128), DMA pieces are 8 rows x 128 B, and
// a slot feeds two multiply steps ks = 0 / 1 (chunks 4 ks .. 4 ks + 3).  Half-step h = 2 t + ks of stage t:
//     group 0: read in phase 2h, multiply in phase 2h + 1;   group 1: one phase later.
// DMA: the pieces of stage t + 2 go into the slot stage t - 1 used (NSTAGE = 3) and are issued from the read phases of
// stage t, a wave's first half of them in its ks = 0 read phase and the rest in its ks = 1 read phase.
//   landing: stage t + 1 was issued during stage t - 1; every wave retires its own pieces of it (counted vmcnt: only the
//            pieces of stage t + 2 may stay in flight) at the end of its ks = 1 read phase of stage t, i.e. before the
//            barrier that ends phase 4t + 2 (group 0) / 4t + 3 (group 1); the first read of stage t + 1 is phase 4t + 4.
//   re-use:  the last reads of stage t - 1 are group 1's ks = 1 reads in phase 4t - 1, retired (lgkmcnt) before the
//            barrier that ends it; the earliest write into that slot is issued in phase 4t.
// Stages past the end of K are still issued (clamped source) so every wait is a constant count.  nk32 = number of
// 32-deep steps (the last stage may hold one step only).
// -------------------------------------------------------------------------------------------------
template <class C, bool SWAP>
DEVINL void big_mainloop64(unsigned char* smem, const char* (&sptr)[C::LPS_LO + 1], const unsigned (&sadv)[C::LPS_LO + 1],
                           const unsigned (&voff)[C::LPS_LO + 1], const int (&dst)[C::LPS_LO + 1], int nk32, int wave,
                           int wm, int wn, int lane, typename C::AccT (&acc)[C::MR][C::NR]) {
  constexpr int MR = C::MR, NR = C::NR, NST = C::NSTAGE, STAGE = C::STAGE_BYTES, D = NST - 1;
  constexpr int LPS_LO = C::LPS_LO, EXTRA = C::EXTRA;
  static_assert(C::KS == 64 && (NST == 2 || NST == 3), "64-deep slots on a 2- or 3-slot ring");
  // NST = 3: the pieces of stage t + 2 are issued from the read phases of stage t, half in ks = 0 and half in ks = 1.
  // NST = 2 (the 288x256 / 256x256 tiles: 68 / 64 KiB per slot): stage t + 1 goes into the slot stage t - 1 used and is
  // issued entirely from the ks = 0 read phases of stage t; its pieces must have landed two phases later.
  constexpr int H0 = (D == 2) ? LPS_LO / 2 : LPS_LO;  // pieces [0, H0) go out in the ks = 0 read phase
  const int grp = wave >> 2;
  const bool extra = EXTRA && wave < EXTRA;
  const int nst = (nk32 + 1) >> 1;  // stages

  auto issue = [&](auto slot_c, auto part_c, bool more) {
    constexpr int SLOT = decltype(slot_c)::value, PART = decltype(part_c)::value;
    unsigned char* base = smem + SLOT * STAGE;
    constexpr bool TAIL_HERE = (D == 2) ? (PART == 1) : (PART == 0);  // which part carries the extra piece
    constexpr int LO = (PART == 0) ? 0 : H0, HI = (PART == 0) ? H0 : LPS_LO;
#pragma unroll
    for (int i = LO; i < HI; ++i) {
      glds16(reinterpret_cast<const bf16_t*>(sptr[i] + voff[i]), base + dst[i]);
      sptr[i] += more ? sadv[i] : 0u;
    }
    if constexpr (TAIL_HERE) {
      if (extra) glds16(reinterpret_cast<const bf16_t*>(sptr[LPS_LO] + voff[LPS_LO]), base + dst[LPS_LO]);
      sptr[LPS_LO] += more ? sadv[LPS_LO] : 0u;
    }
  };
  auto wait_landed = [&]() {  // at most D - 1 later stages of this wave stay in flight
    if constexpr (D == 2) {
      if (extra) wait_vmcnt<LPS_LO + 1>();
      else wait_vmcnt<LPS_LO>();
    } else {
      wait_vmcnt<0>();
    }
  };
  constexpr int MF = C::MF, TILEB = MF * 128;  // bytes of one row tile of the stage image (MF rows x one 128-byte line)
  constexpr int SUBS = (MF == 32) ? 2 : 1;     // MFMA K sub-steps per 32-deep multiply step (32 x 32 x 16: two of 16)
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = typename C::AccT{};

  // fragment addresses.  16 x 16 x 32: row r = lane & 15 of a 16-row tile (2 KiB), chunk 4 ks + (lane >> 4), image position
  // chunk ^ (r & 7).  32 x 32 x 16: row r = lane & 31 of a 32-row tile (4 KiB), chunk 2 sub + (lane >> 5) for K sub-step
  // sub = 0..3 of the slot, image position chunk ^ ((r >> 1) & 7): the 16 lanes of every ds_read_b128 service group
  // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and their upper halves) then fall on 16 distinct 16-byte bank slots.
  unsigned lp[2 * SUBS];
  if constexpr (MF == 32) {
    const int r = lane & 31, hf = lane >> 5, gsw = (r >> 1) & 7;
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) lp[sb] = (unsigned)(r * 128 + (((2 * sb + hf) ^ gsw) << 4));
  } else {
    const int r = lane & 15, cq = lane >> 4;
    lp[0] = (unsigned)(r * 128 + (((0 + cq) ^ (r & 7)) << 4));
    lp[1] = (unsigned)(r * 128 + (((4 + cq) ^ (r & 7)) << 4));
  }
  const unsigned a_base = (unsigned)(wm * MR * TILEB), b_base = (unsigned)(C::A_BYTES + wn * NR * TILEB);

  issue(SlotC<0>{}, SlotC<0>{}, 1 < nst);
  issue(SlotC<0>{}, SlotC<1>{}, 1 < nst);
  if constexpr (D == 2) {
    issue(SlotC<1>{}, SlotC<0>{}, 2 < nst);
    issue(SlotC<1>{}, SlotC<1>{}, 2 < nst);
  }
  wait_landed();
  __builtin_amdgcn_s_barrier();  // stage 0 landed

  bf16x8 af[SUBS][MR], bfr[SUBS][NR];
  auto body = [&](auto slot_c, auto ks_c, int t) {
    constexpr int SLOT = decltype(slot_c)::value, KSI = decltype(ks_c)::value;
    constexpr int NEXT = (SLOT + D) % NST;  // slot of stage t + D (= the slot stage t - 1 used)
    const unsigned char* st = smem + SLOT * STAGE;
#pragma unroll
    for (int sb = 0; sb < SUBS; ++sb) {
#pragma unroll
      for (int j = 0; j < NR; ++j) bfr[sb][j] = *reinterpret_cast<const bf16x8*>(st + b_base + j * TILEB + lp[KSI * SUBS + sb]);
#pragma unroll
      for (int i = 0; i < MR; ++i) af[sb][i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * TILEB + lp[KSI * SUBS + sb]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (D == 2) {
      issue(SlotC<NEXT>{}, SlotC<KSI>{}, t + D + 1 < nst);
    } else if constexpr (KSI == 0) {
      issue(SlotC<NEXT>{}, SlotC<0>{}, t + D + 1 < nst);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (KSI == 1) wait_landed();  // this wave's pieces of stage t + 1 have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int sb = 0; sb < SUBS; ++sb)
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          if constexpr (MF == 32) acc[i][j] = SWAP ? mfma32(bfr[sb][j], af[sb][i], acc[i][j]) : mfma32(af[sb][i], bfr[sb][j], acc[i][j]);
          else acc[i][j] = SWAP ? mfma16(bfr[sb][j], af[sb][i], acc[i][j]) : mfma16(af[sb][i], bfr[sb][j], acc[i][j]);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  };
  // ks = 1 of the last stage when nk32 is odd (K = 800, 2400 ...): that half of the lines is pitch padding and must not be
  // multiplied - the wave only keeps the barrier / DMA / wait pattern of the half-step.
  auto body_tail = [&](auto slot_c, int t) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int NEXT = (SLOT + D) % NST;
    if constexpr (D == 2) issue(SlotC<NEXT>{}, SlotC<1>{}, false);
    wait_landed();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
  };
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one phase behind group 0
#define BIG64_STAGE(SL, T)                                                        \
    if ((T) < nst) {                                                              \
      body(SlotC<SL>{}, SlotC<0>{}, (T));                                         \
      if (2 * (T) + 1 < nk32) body(SlotC<SL>{}, SlotC<1>{}, (T));                 \
      else body_tail(SlotC<SL>{}, (T));                                           \
    }
  for (int t = 0; t < nst; t += NST) {
    BIG64_STAGE(0, t)
    BIG64_STAGE(1, t + 1)
    if constexpr (NST == 3) { BIG64_STAGE(2, t + 2) }
  }
#undef BIG64_STAGE
  if (grp == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();  // the run-ahead stages past the end of K
  __builtin_amdgcn_s_barrier();
}

// XCD-aware re-deal of the 1-D grid: block b runs on XCD b % 8 (observed); each XCD gets a contiguous
// range of logical ids so neighbouring tiles (shared operand panels) hit one private L2.  Bijective.
DEVINL int xcd_logical_id() {
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// exact x / d for x < 65536, d < 65536 with magic = 2^32 / d + 1 (host)
DEVINL unsigned fastdiv(unsigned x, unsigned magic) { return __umulhi(x, magic); }

// -------------------------------------------------------------------------------------------------
// NT kernel: C = A B^T with the fused epilogues of gemm.h.  bf16 / fp32 row outputs leave through a
// wave-private staging area in the (idle) stage ring so that every lane stores 16 bytes of a whole row
// segment (a wave store in the MFMA layout covers 16 rows x 32 / 64 B; under contention that costs 10-20 %
// of the launch).  Staging rows carry one 16-byte pad chunk (conflict-free ds_write_b128, 2-way
// ds_write_b64, linear ds_read_b128).
// -------------------------------------------------------------------------------------------------
// 16-byte global store of an epilogue row chunk (plain store; write-through / non-temporal forms were measured in round 5
// and removed in round 6: profiles/r05_ab_writethrough_epilogue.txt)
template <typename V>
DEVINL void st_out16(void* p, V v) {
  static_assert(sizeof(V) == 16, "16-byte chunk");
  *reinterpret_cast<V*>(p) = v;
}

template <int EPI>
constexpr bool kStagedBf16 = (EPI == EPI_BF16 || EPI == EPI_BIAS_GELU || EPI == EPI_GELU_BWD || EPI == EPI_HEADS);
template <int EPI>
constexpr bool kStagedF32 = (EPI == EPI_F32_BIAS || EPI == EPI_F32_BIAS_RESID);

template <int EPI>
DEVINL void direct_store(const EpiParams& ep, int M, int N, int row, int col0, f32x4 v);

// accumulator rows (MF-row MFMA tiles) per pass of the staged fp32 epilogue: the largest divisor of MR whose 8 wave regions fit
// the idle ring
template <class C>
constexpr int f32_pass_rows() {
  constexpr int MR = C::MR, RT = C::MF, STR = C::WCOLS * 4 + 16;
  return (C::NW * MR * RT * STR <= C::LDS_BYTES)                          ? MR
         : (MR % 2 == 0 && C::NW * (MR / 2) * RT * STR <= C::LDS_BYTES)  ? MR / 2
         : (MR % 3 == 0 && C::NW * (MR / 3) * RT * STR <= C::LDS_BYTES)  ? MR / 3
         : (MR % 4 == 0 && C::NW * (MR / 4) * RT * STR <= C::LDS_BYTES)  ? MR / 4
                                                                        : 1;
}

template <class C, int EPI>
__global__ __launch_bounds__(C::NW * 64, 2 * C::WGS_PER_CU) void big_nt_kernel(const GemmParams p) {
  constexpr int MR = C::MR, NR = C::NR, BM = C::BM, BN = C::BN, NQ = C::NQ, RT = C::MF;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WGN, wn = wave % C::WGN;
  const int lrow_t = lane & (RT - 1);  // this lane's row inside a row tile of the accumulator

  const int tn = (p.N + BN - 1) / BN, tm = (p.M + BM - 1) / BM;
  int m0, n0;
  // split-K: the slices of a tile are neighbouring logical ids (usually the same XCD - not when an XCD's share of the grid is
  // odd, e.g. ids 28 / 29 of 230 - which only matters for speed: the finish is placement-independent)
  const int lid = xcd_logical_id();
  const int z = lid % p.splitk, tile = lid / p.splitk;
  {
    const int id = tile;
    // bands of `band` m-tiles, m fastest inside a band: the ~32 tiles of an XCD form a near-square patch
    const int band = p.band > 0 ? p.band : 4, per = band * tn;  // host: band_for() (near-square XCD patch)
    const int b = id / per, w = id - b * per;
    const int hb = min(band, tm - b * band);
    m0 = (b * band + w % hb) * BM;
    n0 = (w / hb) * BN;
  }
  const int nk_all = p.K / 32;
  int nk_per = (nk_all + p.splitk - 1) / p.splitk;
  if constexpr (C::KS == 64) nk_per = (nk_per + 1) & ~1;  // slices start on 64-deep stage boundaries
  const int k_beg = z * nk_per, nk = min(nk_per, nk_all - k_beg);  // host: every slice has >= 1 stage

  const char* sptr[C::LPS_LO + 1];
  unsigned sadv[C::LPS_LO + 1], voff[C::LPS_LO + 1];
  int dst[C::LPS_LO + 1];
  {
    // 32-deep slots: a piece is 16 rows x 64 B (lane -> row lane >> 2, chunk (lane & 3) ^ ring_g);
    // 64-deep slots: 8 rows x 128 B (lane -> row lane >> 3, chunk (lane & 7) ^ row)
    constexpr int PR = C::PROWS;
    const int lrow = (C::KS == 64) ? (lane >> 3) : (lane >> 2);
    const int lchunk16 = (C::KS == 64) ? ((lane & 7) ^ lrow) : ((lane & 3) ^ ring_g(lrow));
    static_assert(C::MF == 16 || (C::A_PIECES % 4 == 0 && C::B_PIECES % 4 == 0), "32-row tiles = 4 DMA pieces");
#pragma unroll
    for (int i = 0; i < C::LPS_LO + 1; ++i) {
      const int q = (i < C::LPS_LO) ? wave * C::LPS_LO + i : C::NW * C::LPS_LO + wave;  // extras: pieces NW*LPS_LO..
      const int qq = min(q, C::NPIECE - 1);
      // 32 x 32 x 16 image: row r of a 32-row tile keeps chunk c at position c ^ ((r >> 1) & 7); r = (piece & 3) * 8 + lrow
      const int lchunk = (C::MF == 32) ? ((lane & 7) ^ (((qq & 1) << 2) | (lrow >> 1))) : lchunk16;
      if (qq < C::A_PIECES) {  // wave-uniform
        sptr[i] = reinterpret_cast<const char*>(p.A) + (size_t)k_beg * 64;
        voff[i] = ((unsigned)min(m0 + qq * PR + lrow, p.M - 1) * (unsigned)p.lda + lchunk * 8) * 2u;
        dst[i] = qq * 1024;
      } else {
        sptr[i] = reinterpret_cast<const char*>(p.B) + (size_t)k_beg * 64;
        voff[i] = ((unsigned)min(n0 + (qq - C::A_PIECES) * PR + lrow, p.N - 1) * (unsigned)p.ldb + lchunk * 8) * 2u;
        dst[i] = C::A_BYTES + (qq - C::A_PIECES) * 1024;
      }
      sadv[i] = (C::KS == 64) ? 128 : 64;  // bytes along K per stage
    }
  }
  typename C::AccT acc[MR][NR];
  if constexpr (C::KS == 64) big_mainloop64<C, true>(smem, sptr, sadv, voff, dst, nk, wave, wm, wn, lane, acc);
  else big_mainloop<C, false, true>(smem, sptr, sadv, voff, dst, nk, wave, wm, wn, lane, acc);
  // quad jq of accumulator row i: 4 consecutive output columns C::qcol(jq, lane).. of output row i * RT + lrow_t
  // (always_inline: called from a few hundred unrolled sites - out of line, `acc` would live in scratch memory)
  auto quad = [&](int i, int jq) __attribute__((always_inline)) -> f32x4 {
    if constexpr (C::MF == 32) {
      const f32x16& t = acc[i][jq >> 2];
      const int q = (jq & 3) * 4;
      return f32x4{t[q], t[q + 1], t[q + 2], t[q + 3]};
    } else {
      return acc[i][jq];
    }
  };
  auto add_quad = [&](int i, int jq, f32x4 v) __attribute__((always_inline)) {
    if constexpr (C::MF == 32) {
      f32x16& t = acc[i][jq >> 2];
      const int q = (jq & 3) * 4;
      t[q] += v[0]; t[q + 1] += v[1]; t[q + 2] += v[2]; t[q + 3] += v[3];
    } else {
      acc[i][jq] += v;
    }
  };

  if (p.splitk > 1) {
    // In-launch split-K finish (cdna_hip_programming.md 5 "in-launch split-K reduction", write-through form):
    // every slice publishes its fp32 partial tile with sc1 (write-through) 16-byte stores in fragment order
    // (a wave store = 1 KiB contiguous), drains them, and takes a ticket on the tile's arrival counter; the
    // slice that draws the last ticket adds its peers' partials (sc1 loads behind one agent-scope acquire)
    // and runs the epilogue.  No spinning: an early slice exits.  Placement-independent; the counter is
    // returned to zero by the last arriver.
    // (Round 5 added a SYMMETRIC 2-way finish - both slices stay, wait for each other on the counter and each run half the
    //  epilogue: 0.02 ms per step, inside the box spread, for a spin-wait whose safety needs every workgroup of the launch to
    //  be resident, which the engine cannot guarantee against other handles / processes / RCCL kernels.  Removed in round 6:
    //  profiles/r05_ab_symmetric_splitk.txt, git history.)
    constexpr unsigned WG_BYTES = (unsigned)BM * BN * 4;
    char* tile_slabs = reinterpret_cast<char*>(p.sk_slab) + (size_t)tile * p.splitk * WG_BYTES;
    const unsigned lane_off = (unsigned)(wave * MR * NQ * 1024 + lane * 16);
    {
      const __amdgpu_buffer_rsrc_t mine =
          __builtin_amdgcn_make_buffer_rsrc(tile_slabs + (size_t)z * WG_BYTES, 0, WG_BYTES, 0x00020000);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NQ; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, quad(i, j)), mine,
                                                 lane_off + (i * NQ + j) * 1024, 0, /*sc1*/ 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(smem);  // the ring is idle: no second __shared__ object
    if (tid == 0)
      *flag = __hip_atomic_fetch_add(p.sk_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = *flag;
    if (ticket != (unsigned)(p.splitk - 1)) return;
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(p.sk_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int zz = 0; zz < p.splitk; ++zz) {
      if (zz == z) continue;
      const __amdgpu_buffer_rsrc_t peer =
          __builtin_amdgcn_make_buffer_rsrc(tile_slabs + (size_t)zz * WG_BYTES, 0, WG_BYTES, 0x00020000);
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        u32x4 t[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j)
          t[j] = __builtin_amdgcn_raw_buffer_load_b128(peer, lane_off + (i * NQ + j) * 1024, 0, /*sc1*/ 16);
#pragma unroll
        for (int j = 0; j < NQ; ++j) add_quad(i, j, __builtin_bit_cast(f32x4, t[j]));
      }
    }
    __syncthreads();  // flag word read by everyone before the ring is reused as epilogue staging
  }

  const EpiParams& ep = p.ep;
  const int wrow0 = m0 + wm * C::WROWS, wcol0 = n0 + wn * C::WCOLS;
  if constexpr (kStagedBf16<EPI>) {
    constexpr int NOUT = (EPI == EPI_BIAS_GELU) ? 2 : 1;
    constexpr int ROWB = C::WCOLS * 2, STR = ROWB + 16, CPR = ROWB / 16;  // bf16 staging row, chunks per row
    // rows per pass: largest divisor CH of MR whose 8 wave regions fit the ring
    constexpr int CH = (C::NW * NOUT * MR * RT * STR <= C::LDS_BYTES)                             ? MR
                       : (MR % 2 == 0 && C::NW * NOUT * (MR / 2) * RT * STR <= C::LDS_BYTES)     ? MR / 2
                       : (MR % 3 == 0 && C::NW * NOUT * (MR / 3) * RT * STR <= C::LDS_BYTES)     ? MR / 3
                       : (MR % 4 == 0 && C::NW * NOUT * (MR / 4) * RT * STR <= C::LDS_BYTES)     ? MR / 4
                                                                                              : 1;
    static_assert(C::NW * NOUT * CH * RT * STR <= C::LDS_BYTES, "one pass of the bf16 staging fits the ring");
    constexpr int REG = CH * RT * STR;
    constexpr int TOT = CH * RT * CPR, IT = (TOT + 63) / 64;  // 16-byte chunks of one pass, per-lane trips
    bool aligned = !(p.N & 7) && (NOUT == 1 || !(ep.ldo1 & 7)) && (EPI != EPI_GELU_BWD || !(ep.ldp & 7));
    if constexpr (EPI == EPI_HEADS) aligned = aligned && !(ep.dh & 7) && !(ep.dhp & 7);
    else aligned = aligned && !(ep.ldo0 & 7);
    if (aligned) {
      unsigned char* stg = smem + wave * (NOUT * REG);
      bf16_t* h0 = ep.hrow[0];
      bf16_t* h1 = ep.hrow[1];
      bf16_t* h2 = ep.hrow[2];
#pragma unroll
      for (int c = 0; c < MR / CH; ++c) {
        const int prow0 = wrow0 + c * CH * RT;
        if constexpr (EPI == EPI_GELU_BWD) {
          // the saved pre-activation comes in the way the result goes out: whole row segments into the
          // staging image, then each lane picks its 4 values at the offset it will overwrite
          bf16x8 pv[IT];
#pragma unroll
          for (int it = 0; it < IT; ++it) {
            const int q = it * 64 + lane;
            const int lr = q / CPR, ch = q - lr * CPR;
            const int row = min(prow0 + lr, p.M - 1), col = wcol0 + ch * 8;
            pv[it] = bf16x8{};
            if (q < TOT && col < p.N) pv[it] = *reinterpret_cast<const bf16x8*>(ep.pre + (size_t)row * ep.ldp + col);
          }
#pragma unroll
          for (int it = 0; it < IT; ++it) {
            const int q = it * 64 + lane;
            const int lr = q / CPR, ch = q - lr * CPR;
            if (q < TOT) *reinterpret_cast<bf16x8*>(stg + lr * STR + ch * 16) = pv[it];
          }
        }
#pragma unroll
        for (int ii = 0; ii < CH; ++ii) {
          const int i = c * CH + ii;
          const int lr = ii * RT + lrow_t;
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            const int lc = C::qcol(j, lane);
            const int col = wcol0 + lc;
            const f32x4 v = quad(i, j);
            const int off = lr * STR + lc * 2;
            bf16x4 o0, o1;
            if constexpr (EPI == EPI_BF16 || EPI == EPI_HEADS) {
              o0 = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            } else if constexpr (EPI == EPI_BIAS_GELU) {
              float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
              if (col < p.N) b = *reinterpret_cast<const float4*>(ep.bias + col);
              const float x0 = v[0] + b.x, x1 = v[1] + b.y, x2 = v[2] + b.z, x3 = v[3] + b.w;
              o0 = bf16x4{(bf16_t)x0, (bf16_t)x1, (bf16_t)x2, (bf16_t)x3};
              o1 = bf16x4{(bf16_t)gelu_tanh(x0), (bf16_t)gelu_tanh(x1), (bf16_t)gelu_tanh(x2), (bf16_t)gelu_tanh(x3)};
            } else {  // EPI_GELU_BWD
              const bf16x4 pv = *reinterpret_cast<const bf16x4*>(stg + off);
              o0 = bf16x4{(bf16_t)(v[0] * gelu_tanh_grad((float)pv[0])), (bf16_t)(v[1] * gelu_tanh_grad((float)pv[1])),
                          (bf16_t)(v[2] * gelu_tanh_grad((float)pv[2])), (bf16_t)(v[3] * gelu_tanh_grad((float)pv[3]))};
            }
            *reinterpret_cast<bf16x4*>(stg + off) = o0;
            if constexpr (NOUT == 2) *reinterpret_cast<bf16x4*>(stg + REG + off) = o1;
          }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const int q = it * 64 + lane;
          const int lr = q / CPR, ch = q - lr * CPR;
          const int row = prow0 + lr, col = wcol0 + ch * 8;
          if (q < TOT && row < p.M && col < p.N) {
            const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(stg + lr * STR + ch * 16);
            if constexpr (EPI == EPI_HEADS) {
              // column -> (which, h, d), row -> (b, t); an 8-column chunk never straddles a head (dh % 8 == 0)
              const unsigned which = fastdiv(col, ep.mg_hid), rem = col - which * ep.hid;
              const unsigned h = fastdiv(rem, ep.mg_dh), d = rem - h * ep.dh;
              const unsigned b = fastdiv(row, ep.mg_ntok), t = row - b * ep.n_tok;
              bf16_t* hr = (which == 0) ? h0 : (which == 1) ? h1 : h2;
              if (hr) st_out16(hr + ((size_t)(b * ep.heads + h) * ep.n_pad + t) * ep.dhp + d, v0);
            } else {
              if (NOUT == 1 || ep.out0) st_out16((bf16_t*)ep.out0 + (size_t)row * ep.ldo0 + col, v0);
              if constexpr (NOUT == 2) {
                const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(stg + REG + lr * STR + ch * 16);
                st_out16((bf16_t*)ep.out1 + (size_t)row * ep.ldo1 + col, v1);
              }
            }
          }
        }
      }
      return;
    }
  }
  if constexpr (kStagedF32<EPI>) {
    constexpr int ROWB = C::WCOLS * 4, STR = ROWB + 16, CPR = ROWB / 16;
    constexpr int CH = f32_pass_rows<C>();
    static_assert(C::NW * CH * RT * STR <= C::LDS_BYTES, "one pass of the fp32 staging fits the ring");
    constexpr int REG = CH * RT * STR;
    constexpr int TOT = CH * RT * CPR, IT = (TOT + 63) / 64;
    const bool aligned = !(p.N & 3) && !(ep.ldo0 & 3) && (EPI != EPI_F32_BIAS_RESID || !(ep.ldr & 3));
    if (aligned) {
      unsigned char* stg = smem + wave * REG;
#pragma unroll
      for (int c = 0; c < MR / CH; ++c) {
        const int prow0 = wrow0 + c * CH * RT;
        // the fp32 residual rows of this pass are requested first, whole row segments per wave load; their
        // latency hides behind the accumulator -> LDS re-shape below
        float4 rv[IT];
        if constexpr (EPI == EPI_F32_BIAS_RESID) {
#pragma unroll
          for (int it = 0; it < IT; ++it) {
            const int q = it * 64 + lane;
            const int lr = q / CPR, ch = q - lr * CPR;
            const int row = prow0 + lr, col = wcol0 + ch * 4;
            rv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < TOT && row < p.M && col < p.N)
              rv[it] = *reinterpret_cast<const float4*>(ep.resid + (size_t)row * ep.ldr + col);
          }
        }
#pragma unroll
        for (int ii = 0; ii < CH; ++ii) {
          const int lr = ii * RT + lrow_t;
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            const f32x4 v = quad(c * CH + ii, j);
            *reinterpret_cast<float4*>(stg + lr * STR + C::qcol(j, lane) * 4) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const int q = it * 64 + lane;
          const int lr = q / CPR, ch = q - lr * CPR;
          const int row = prow0 + lr, col = wcol0 + ch * 4;
          if (q < TOT && row < p.M && col < p.N) {
            float4 v = *reinterpret_cast<const float4*>(stg + lr * STR + ch * 16);
            if (ep.bias) {
              const float4 b = *reinterpret_cast<const float4*>(ep.bias + col);
              v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            if constexpr (EPI == EPI_F32_BIAS_RESID) {
              v.x += rv[it].x; v.y += rv[it].y; v.z += rv[it].z; v.w += rv[it].w;
            }
            st_out16((float*)ep.out0 + (size_t)row * ep.ldo0 + col, v);
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < MR; ++i) {
#pragma unroll
    for (int j = 0; j < NQ; ++j)
      direct_store<EPI>(ep, p.M, p.N, wrow0 + i * RT + lrow_t, wcol0 + C::qcol(j, lane), quad(i, j));
  }
}

// One lane's share of the un-staged epilogue: output row `row`, columns col0..col0+3 (col0 % 4 == 0).
template <int EPI>
DEVINL void direct_store(const EpiParams& ep, int M, int N, int row, int col0, f32x4 v) {
  if (row >= M || col0 >= N) return;
  if constexpr (EPI == EPI_BF16) {
    bf16_t* o = (bf16_t*)ep.out0 + (size_t)row * ep.ldo0 + col0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col0 + r < N) o[r] = (bf16_t)v[r];
  } else if constexpr (EPI == EPI_F32_BIAS || EPI == EPI_F32_BIAS_POS || EPI == EPI_F32_BIAS_RESID) {
    float* o = (float*)ep.out0 + (size_t)row * ep.ldo0 + col0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col0 + r < N) {
        float x = v[r] + (ep.bias ? ep.bias[col0 + r] : 0.f);
        if constexpr (EPI == EPI_F32_BIAS_POS) x += ep.pos[(size_t)(row % ep.seq) * N + col0 + r];
        if constexpr (EPI == EPI_F32_BIAS_RESID) x += ep.resid[(size_t)row * ep.ldr + col0 + r];
        o[r] = x;
      }
  } else if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col0 + r < N) {
        const float pre = v[r] + ep.bias[col0 + r];
        if (ep.out0) ((bf16_t*)ep.out0)[(size_t)row * ep.ldo0 + col0 + r] = (bf16_t)pre;
        ((bf16_t*)ep.out1)[(size_t)row * ep.ldo1 + col0 + r] = (bf16_t)gelu_tanh(pre);
      }
  } else if constexpr (EPI == EPI_GELU_BWD) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col0 + r < N)
        ((bf16_t*)ep.out0)[(size_t)row * ep.ldo0 + col0 + r] =
            (bf16_t)(v[r] * gelu_tanh_grad((float)ep.pre[(size_t)row * ep.ldp + col0 + r]));
  } else if constexpr (EPI == EPI_HEADS) {
    const int which = col0 / ep.hid, rem = col0 - which * ep.hid;
    const int h = rem / ep.dh, d = rem - h * ep.dh;
    const int b = row / ep.n_tok, t = row - b * ep.n_tok;
    bf16_t* hr = (which == 0) ? ep.hrow[0] : (which == 1) ? ep.hrow[1] : ep.hrow[2];
    if (hr) {
      bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(hr + ((size_t)(b * ep.heads + h) * ep.n_pad + t) * ep.dhp + d) = pk;
    }
  } else if constexpr (EPI == EPI_F32_BF16) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col0 + r < N) {
        ((float*)ep.out0)[(size_t)row * ep.ldo0 + col0 + r] = v[r];
        ((bf16_t*)ep.out1)[(size_t)row * ep.ldo1 + col0 + r] = (bf16_t)v[r];
      }
  }
}

// -------------------------------------------------------------------------------------------------
// Grouped TN kernel (weight gradients): out[m][n] += sum_k A[k][m] B[k][n] over the WHOLE K range, fp32.
// Logical block id -> (problem, m-tile, n-tile), m fastest: with 5 m-tiles per 800-row side the ~24 tiles
// an XCD owns form a near-square patch (5 x 5 operand panels per K step instead of 24 + 24).
// -------------------------------------------------------------------------------------------------
// Fused optimizer epilogue of the grouped TN kernel (TnGroup::adam, gemm.h).  On entry acc holds the FINAL gradient of
// the wave's sub-tile (whole-K launch, overwrite semantics) in the lane layout of the branch that produced it:
//   !TRANS (swapped roles): lane = row m = mrow[i] + (lane & 15), 4 consecutive n = ncol[j] + (lane >> 4) * 4, out[m][n]
//    TRANS                : lane = column n = ncol[j] + (lane & 15), 4 consecutive m = mrow[i] + (lane >> 4) * 4, out[n][m]
// Per accumulator row i the wave requests p / m / v of all NR tiles (one row ahead of the arithmetic), updates them with
// adam1 (common.h: the optimizer pass's own arithmetic), stores them back and writes the bf16 shadow that has out's
// orientation straight from the registers (8 bytes per lane).  The TRANSPOSED shadow goes through a wave-private LDS tile
// (the ring is idle after the main loop's closing vmcnt(0) + barrier): bf16 elements are scattered in transposed order,
// then read back as 16-byte row chunks and stored as whole row segments.  No barrier: the tile belongs to one wave.
template <class C, bool TRANS>
DEVINL void tn_adam_epilogue(unsigned char* smem, const TnProblem& pr, float lr_t, float b1, float b2, float eps,
                             f32x4 (&acc)[C::MR][C::NR], const int (&mrow)[C::MR], const int (&ncol)[C::NR], int m0,
                             int n0, int wave, int wm, int wn, int lane) {
  constexpr int MR = C::MR, NR = C::NR;
  constexpr int TROWS = TRANS ? MR * 16 : NR * 16;  // rows of the transposed shadow this wave owns
  constexpr int TCOLS = TRANS ? NR * 16 : MR * 16;  // contiguous elements per row
  constexpr int PITCH = TCOLS + 8;                  // bf16 elements: 16-byte aligned rows, an odd number of 16-byte slots
  static_assert(C::NW * TROWS * PITCH * 2 <= C::LDS_BYTES, "transposition tiles fit the idle ring");
  bf16_t* tb = reinterpret_cast<bf16_t*>(smem) + wave * (TROWS * PITCH);
  const int l15 = lane & 15, l4 = lane >> 4;
  f32x4 pv[2][NR], mv[2][NR], vv[2][NR];
  auto elem = [&](int i, int j, bool& ok) -> size_t {  // element offset of this lane's 4 values of tile (i, j)
    if constexpr (!TRANS) {
      const int m = mrow[i] + l15, n = ncol[j] + l4 * 4;
      ok = m < pr.M && n + 3 < pr.N;
      return (size_t)m * pr.ldo + n;
    } else {
      const int m = mrow[i] + l4 * 4, n = ncol[j] + l15;
      ok = n < pr.N && m + 3 < pr.M;
      return (size_t)n * pr.ldo + m;
    }
  };
  auto request = [&](int i, int b) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      bool ok;
      const size_t o = elem(i, j, ok);
      if (ok) {
        pv[b][j] = *reinterpret_cast<const f32x4*>(pr.p + o);
        mv[b][j] = *reinterpret_cast<const f32x4*>(pr.m1 + o);
        vv[b][j] = *reinterpret_cast<const f32x4*>(pr.v + o);
      }
    }
  };
  request(0, 0);
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const int b = i & 1;
    if (i + 1 < MR) request(i + 1, b ^ 1);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      bool ok;
      const size_t o = elem(i, j, ok);
      if (!ok) continue;
      f32x4 pn = pv[b][j], mn = mv[b][j], vn = vv[b][j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pj = pn[r], mj = mn[r], vj = vn[r];
        adam1(pj, mj, vj, acc[i][j][r], lr_t, b1, b2, eps);
        pn[r] = pj; mn[r] = mj; vn[r] = vj;
      }
      *reinterpret_cast<f32x4*>(pr.p + o) = pn;
      *reinterpret_cast<f32x4*>(pr.m1 + o) = mn;
      *reinterpret_cast<f32x4*>(pr.v + o) = vn;
      const bf16x4 w16 = {(bf16_t)pn[0], (bf16_t)pn[1], (bf16_t)pn[2], (bf16_t)pn[3]};
      if constexpr (!TRANS) {
        const int m = mrow[i] + l15, n = ncol[j] + l4 * 4;
        *reinterpret_cast<bf16x4*>(pr.sd + (size_t)m * pr.ldsd + n) = w16;
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[(j * 16 + l4 * 4 + r) * PITCH + i * 16 + l15] = w16[r];
      } else {
        const int m = mrow[i] + l4 * 4, n = ncol[j] + l15;
        *reinterpret_cast<bf16x4*>(pr.sd + (size_t)n * pr.ldsd + m) = w16;
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[(i * 16 + l4 * 4 + r) * PITCH + j * 16 + l15] = w16[r];
      }
    }
  }
  // transposed shadow: 16-byte chunks of the wave's TROWS x TCOLS tile -> st[row][col .. col + 7]
  constexpr int CPR = TCOLS / 8, ITEMS = TROWS * CPR;
  static_assert(ITEMS % 64 == 0, "whole wave trips");
#pragma unroll
  for (int it = 0; it < ITEMS / 64; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / CPR, c = idx - row * CPR;
    // local 16-wide unit of the row / of the chunk -> global index (the units of a wave are not contiguous: TnImg::unit_of)
    int gm, gn;
    if constexpr (!TRANS) {
      gn = n0 + TnImg<C::BN>::template unit_of<C::WGN, NR>(wn, row >> 4) * 16 + (row & 15);
      gm = m0 + TnImg<C::BM>::template unit_of<C::WGM, MR>(wm, c >> 1) * 16 + (c & 1) * 8;
      if (gn < pr.N && gm + 7 < pr.M)
        *reinterpret_cast<u32x4*>(pr.st + (size_t)gn * pr.ldst + gm) = *reinterpret_cast<const u32x4*>(tb + row * PITCH + c * 8);
    } else {
      gm = m0 + TnImg<C::BM>::template unit_of<C::WGM, MR>(wm, row >> 4) * 16 + (row & 15);
      gn = n0 + TnImg<C::BN>::template unit_of<C::WGN, NR>(wn, c >> 1) * 16 + (c & 1) * 8;
      if (gm < pr.M && gn + 7 < pr.N)
        *reinterpret_cast<u32x4*>(pr.st + (size_t)gm * pr.ldst + gn) = *reinterpret_cast<const u32x4*>(tb + row * PITCH + c * 8);
    }
  }
}

template <class C, int PF, bool ADAM = false>
__global__ __launch_bounds__(C::NW * 64, 1) void big_tn_kernel(const TnGroup g) {
  constexpr int MR = C::MR, NR = C::NR, BM = C::BM, BN = C::BN;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WGN, wn = wave % C::WGN;

  const int id = g.tile0 + xcd_logical_id();
  int pi = 0;
#pragma unroll
  for (int i = 1; i < TN_GROUP_MAX; ++i)
    if (i < g.n && id >= g.p[i].tile_begin) pi = i;
  // scalar copies (no dynamic indexing of the by-value kernarg struct: that would spill it to scratch)
  TnProblem pr = g.p[0];
#pragma unroll
  for (int i = 1; i < TN_GROUP_MAX; ++i)
    if (pi == i) pr = g.p[i];
  const int local = id - pr.tile_begin;
  const int tmi = local % pr.tiles_m, tni = local / pr.tiles_m;
  const int m0 = tmi * BM, n0 = tni * BN;
  const int nk = g.K / 32;

  const char* sptr[C::LPS_LO + 1];
  unsigned sadv[C::LPS_LO + 1], voff[C::LPS_LO + 1];
  int dst[C::LPS_LO + 1];
#pragma unroll
  for (int i = 0; i < C::LPS_LO + 1; ++i) {
    const int q = (i < C::LPS_LO) ? wave * C::LPS_LO + i : C::NW * C::LPS_LO + wave;
    const int qq = min(q, C::NPIECE - 1);
    int k, col;
    if (qq < C::A_PIECES) {  // wave-uniform
      TnImg<BM>::piece_src(qq, lane, k, col);
      sptr[i] = reinterpret_cast<const char*>(pr.A);
      voff[i] = ((unsigned)k * (unsigned)pr.lda + (unsigned)min(m0 + col, pr.lda - 8)) * 2u;
      dst[i] = qq * 1024;
      sadv[i] = 64u * (unsigned)pr.lda;  // 32 rows of lda bf16
    } else {
      TnImg<BN>::piece_src(qq - C::A_PIECES, lane, k, col);
      sptr[i] = reinterpret_cast<const char*>(pr.B);
      voff[i] = ((unsigned)k * (unsigned)pr.ldb + (unsigned)min(n0 + col, pr.ldb - 8)) * 2u;
      dst[i] = C::A_BYTES + (qq - C::A_PIECES) * 1024;
      sadv[i] = 64u * (unsigned)pr.ldb;
    }
  }
  f32x4 acc[MR][NR];
  // output rows / columns of this wave's MFMA tiles (unit order of the TN image: main sub-images, then tail)
  int mrow[MR], ncol[NR];
#pragma unroll
  for (int i = 0; i < MR; ++i) mrow[i] = m0 + TnImg<BM>::template unit_of<C::WGM, MR>(wm, i) * 16;
#pragma unroll
  for (int j = 0; j < NR; ++j) ncol[j] = n0 + TnImg<BN>::template unit_of<C::WGN, NR>(wn, j) * 16;
  // column sums of an operand (TnProblem::csum): the staggered loop only
  constexpr int CSN = MR > NR ? MR : NR;
  f32x4 cs[CSN];
#pragma unroll
  for (int x = 0; x < CSN; ++x) cs[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!pr.trans_out) {
    const bool do_cs = (PF == 0) && pr.csum != nullptr && tmi == 0 && wm == 0;
    if constexpr (PF) tn_mainloop_pf<C, true, PF >= 2, (PF > 2 ? PF - 2 : 0)>(smem, sptr, sadv, voff, dst, nk, wave, wm, wn, lane, acc);
    else big_mainloop<C, true, true, true>(smem, sptr, sadv, voff, dst, nk, wave, wm, wn, lane, acc, do_cs, cs);
    if (do_cs && (lane & 15) == 0) {  // cs[j][r] = sum_k B[k][ncol[j] + (lane >> 4) * 4 + r], replicated over lane & 15
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = ncol[j] + (lane >> 4) * 4 + r;
          if (n < pr.N) atomicAdd(pr.csum + n, cs[j][r]);
        }
    }
    if constexpr (ADAM) {
      tn_adam_epilogue<C, false>(smem, pr, g.lr_t, g.b1, g.b2, g.eps, acc, mrow, ncol, m0, n0, wave, wm, wn, lane);
      return;
    }
    // lane: row m = .. + (lane&15), 4 consecutive n.  All old values are requested before the first store.
    float4 old[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int m = mrow[i] + (lane & 15);
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int n = ncol[j] + (lane >> 4) * 4;
        old[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!g.overwrite && m < pr.M && n + 3 < pr.N)
          old[i][j] = *reinterpret_cast<const float4*>(pr.out + (size_t)m * pr.ldo + n);
      }
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int m = mrow[i] + (lane & 15);
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int n = ncol[j] + (lane >> 4) * 4;
        if (m < pr.M && n + 3 < pr.N)
          *reinterpret_cast<float4*>(pr.out + (size_t)m * pr.ldo + n) =
              make_float4(old[i][j].x + acc[i][j][0], old[i][j].y + acc[i][j][1], old[i][j].z + acc[i][j][2],
                          old[i][j].w + acc[i][j][3]);
      }
    }
  } else {
    const bool do_cs = (PF == 0) && pr.csum != nullptr && tni == 0 && wn == 0;
    if constexpr (PF) tn_mainloop_pf<C, false, PF >= 2, (PF > 2 ? PF - 2 : 0)>(smem, sptr, sadv, voff, dst, nk, wave, wm, wn, lane, acc);
    else big_mainloop<C, true, false, true>(smem, sptr, sadv, voff, dst, nk, wave, wm, wn, lane, acc, do_cs, cs);
    if (do_cs && (lane & 15) == 0) {  // cs[i][r] = sum_k A[k][mrow[i] + (lane >> 4) * 4 + r]
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mrow[i] + (lane >> 4) * 4 + r;
          if (m < pr.M) atomicAdd(pr.csum + m, cs[i][r]);
        }
    }
    if constexpr (ADAM) {
      tn_adam_epilogue<C, true>(smem, pr, g.lr_t, g.b1, g.b2, g.eps, acc, mrow, ncol, m0, n0, wave, wm, wn, lane);
      return;
    }
    // un-swapped roles: lane holds 4 consecutive m of column n = .. + (lane&15); out is [n][m]
    float4 old[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int m = mrow[i] + (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int n = ncol[j] + (lane & 15);
        old[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!g.overwrite && n < pr.N && m + 3 < pr.M)
          old[i][j] = *reinterpret_cast<const float4*>(pr.out + (size_t)n * pr.ldo + m);
      }
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int m = mrow[i] + (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int n = ncol[j] + (lane & 15);
        if (n < pr.N && m + 3 < pr.M)
          *reinterpret_cast<float4*>(pr.out + (size_t)n * pr.ldo + m) =
              make_float4(old[i][j].x + acc[i][j][0], old[i][j].y + acc[i][j][1], old[i][j].z + acc[i][j][2],
                          old[i][j].w + acc[i][j][3]);
      }
    }
  }
}

// Skinny-M GEMMs (gemm.h GemmParams::skinny_acc): element-wise epilogue over the fp32 accumulator the split-K
// atomics filled; every element is read once, handed to the fused epilogue and written back as zero.
template <int EPI>
__global__ __launch_bounds__(256) void skinny_epilogue_kernel(float* __restrict__ acc, int ldacc, const GemmParams p) {
  const int c4 = (p.N + 3) >> 2;
  const int total = p.M * c4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / c4, col0 = (i - row * c4) * 4;
    float4* a = reinterpret_cast<float4*>(acc + (size_t)row * ldacc + col0);
    const float4 v = *a;
    *a = make_float4(0.f, 0.f, 0.f, 0.f);
    direct_store<EPI>(p.ep, p.M, p.N, row, col0, f32x4{v.x, v.y, v.z, v.w});
  }
}

// ... fused with the LayerNorm that follows the residual add (EpiParams::ln_*): one wave per output row,
//   x = acc + bias + resid (stored fp32, same operation order as the element-wise pass), acc <- 0,
//   h = (x - mean) * rstd * gamma + beta as bf16 (the arithmetic of rowops.hip ln_fwd_kernel) + the row statistics.
template <int MAXV>
__global__ __launch_bounds__(256) void skinny_epilogue_ln_kernel(float* __restrict__ acc, int ldacc, const GemmParams p) {
  const EpiParams& ep = p.ep;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int nv = p.N >> 2;
  float4* ar = reinterpret_cast<float4*>(acc + (size_t)row * ldacc);
  const float4* rr = reinterpret_cast<const float4*>(ep.resid + (size_t)row * ep.ldr);
  const float4* b4 = reinterpret_cast<const float4*>(ep.bias);
  float4* xo = reinterpret_cast<float4*>((float*)ep.out0 + (size_t)row * ep.ldo0);
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float4 a = ar[idx], b = b4[idx], r = rr[idx];
      ar[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
      v[i] = make_float4((a.x + b.x) + r.x, (a.y + b.y) + r.y, (a.z + b.z) + r.z, (a.w + b.w) + r.w);
      xo[idx] = v[i];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mu = wave_sum(s) / (float)p.N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)p.N + ep.ln_eps);
  if (lane == 0) {
    ep.ln_mean[row] = mu;
    ep.ln_rstd[row] = rs;
  }
  const float4* g4 = reinterpret_cast<const float4*>(ep.ln_g);
  const float4* be4 = reinterpret_cast<const float4*>(ep.ln_b);
  bf16x4* hr = reinterpret_cast<bf16x4*>(ep.ln_h + (size_t)row * ep.ln_ldh);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float4 g = g4[idx], b = be4[idx];
      bf16x4 o = {(bf16_t)((v[i].x - mu) * rs * g.x + b.x), (bf16_t)((v[i].y - mu) * rs * g.y + b.y),
                  (bf16_t)((v[i].z - mu) * rs * g.z + b.z), (bf16_t)((v[i].w - mu) * rs * g.w + b.w)};
      hr[idx] = o;
    }
  }
}

template <typename K>
int allow_lds(K kernel, int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             bytes) == hipSuccess
             ? 0
             : -20;
}

template <class C, int EPI>
int launch_big_nt_cfg(const GemmParams& p, hipStream_t s) {
  static bool once = false;
  if (!once) {
    if (int rc = allow_lds(big_nt_kernel<C, EPI>, C::LDS_BYTES)) return rc;
    once = true;
  }
  const int tiles = ((p.M + C::BM - 1) / C::BM) * ((p.N + C::BN - 1) / C::BN);
  if (p.splitk > 1) {
    if (!p.sk_slab || !p.sk_cnt || tiles > kSplitKCounters ||
        (size_t)tiles * p.splitk * C::BM * C::BN * 4 > kSplitKSlabBytes || p.K / 32 / p.splitk < 1)
      return -9;
    if (C::KS == 64) {  // slices are rounded up to whole 64-deep stages: the last one must not be empty
      const int nk_all = p.K / 32, per = (((nk_all + p.splitk - 1) / p.splitk) + 1) & ~1;
      if ((p.splitk - 1) * per >= nk_all) return -9;
    }
  }
  FACT_LAUNCH((big_nt_kernel<C, EPI>), dim3(tiles * p.splitk), dim3(C::NW * 64), C::LDS_BYTES, s, p);
  return 0;
}

using Cfg288x256 = BigCfg<2, 9, 4, 4>;
using Cfg256x256 = BigCfg<2, 8, 4, 4>;
using Cfg256x160 = BigCfg<4, 4, 2, 5>;
using Cfg160x256 = BigCfg<2, 5, 4, 4>;
using Cfg160x256r6 = BigCfg<2, 5, 4, 4, 6>;  // 6-slot LDS-DMA ring (160 KiB) for the fragment-prefetch TN loop (tn_loop = 6)
// 2 workgroups per CU (3-deep ring, 72 KiB; 64x64 wave tile -> <= 128 VGPRs): the epilogue of one workgroup (bias /
// GELU math, staging, 35-70 MB of stores for the K = 800 GEMMs) runs under the main loop of the other
using Cfg256x128 = BigCfg<4, 4, 2, 4, 3, 2>;
// 256x160 on 64-deep ring slots (3 x 52 KiB): whole 128-byte lines per DMA piece (big_mainloop64)
using Cfg256x160k64 = BigCfg<4, 4, 2, 5, 3, 1, 64>;
using Cfg288x256k64 = BigCfg<2, 9, 4, 4, 2, 1, 64>;  // 2 x 68 KiB
using Cfg256x256k64 = BigCfg<2, 8, 4, 4, 2, 1, 64>;  // 2 x 64 KiB
// 192x160 (wave tile 48 x 80) on 3 x 44 KiB slots: 5760 = 30 x 192 -> 150 tiles where 256x160 gives 115; the two whole-K
// N = 800 dgrads of the backward chain run ONE round beside the 95-workgroup wgrad launch either way, so 25 % less work per
// tile is 15-20 % less kernel (13 % more operand bytes per FLOP)
using Cfg192x160k64 = BigCfg<4, 3, 2, 5, 3, 1, 64>;
// 128x160 (wave tile 32 x 80), 3 x 18 KiB slots, two workgroups per CU: 45 x 5 = 225 tiles for M = 5760, N = 800 - the fp32 +
// residual epilogue of the K = 800 out-projection then streams from 225 CUs instead of 161 (nt variant 21; default for the
// short-K N = 800 GEMMs since round 5, gemm.hip g_tile128x160)
using Cfg128x160 = BigCfg<4, 2, 2, 5, 3, 2>;
// Round 6: v_mfma_f32_32x32x16_bf16 tiles (BigCfg::MF = 32; MR / NR count 32 x 32 tiles), two 64-deep ring slots.
//   256x256: wave grid 2 x 4, wave tile 128 x 64 = 4 x 2 tiles (the 16 x 16 config's geometry, for like-for-like A/Bs)
//   384x192: wave grid 4 x 2, wave tile  96 x 96 = 3 x 3 tiles: M = 5760 = 15 x 384 and N = 3072 = 16 x 192 -> 240 tiles, the
//            count (and area) of 288x256, which has no 32-row decomposition on 8 waves (288 = 9 x 32); 6 fragment reads per 9
//            MFMAs of 32 x 32 x 16 (288x256: 13 per 36 of 16 x 16 x 32 = 6.5 per 9), 72 KiB per slot (68)
using Cfg256x256m32 = BigCfg<2, 4, 4, 2, 2, 1, 64, 32>;
using Cfg384x192m32 = BigCfg<4, 3, 2, 3, 2, 1, 64, 32>;
// (the same tile for the LONG-K N = 800 GEMMs, whole K, 225 workgroups: FFN2 + residual 44.9 us against 40.3 us for 256x160
//  with the symmetric split-K; whole-K dgrads 40.2 vs 42.3 us; on two 64-deep slots 1-4 us slower still -
//  profiles/r05_tile128x160_long_k.txt.  Not used there.)

template <int EPI>
int launch_big_nt_epi(int cfg, const GemmParams& p, hipStream_t s) {
  switch (cfg) {
    case BIG_288x256: return launch_big_nt_cfg<Cfg288x256, EPI>(p, s);
    case BIG_256x256: return launch_big_nt_cfg<Cfg256x256, EPI>(p, s);
    case BIG_256x160: return launch_big_nt_cfg<Cfg256x160, EPI>(p, s);
    case BIG_256x128: return launch_big_nt_cfg<Cfg256x128, EPI>(p, s);
    case BIG_256x160_K64: return launch_big_nt_cfg<Cfg256x160k64, EPI>(p, s);
    case BIG_288x256_K64: return launch_big_nt_cfg<Cfg288x256k64, EPI>(p, s);
    case BIG_256x256_K64: return launch_big_nt_cfg<Cfg256x256k64, EPI>(p, s);
    case BIG_256x256_M32: return launch_big_nt_cfg<Cfg256x256m32, EPI>(p, s);
    case BIG_384x192_M32: return launch_big_nt_cfg<Cfg384x192m32, EPI>(p, s);
    case BIG_192x160_K64:
      if constexpr (EPI == EPI_BF16) return launch_big_nt_cfg<Cfg192x160k64, EPI>(p, s);  // the dgrad epilogue only
      else return -7;
    case BIG_128x160:
      if constexpr (EPI == EPI_F32_BIAS_RESID || EPI == EPI_HEADS || EPI == EPI_BF16) return launch_big_nt_cfg<Cfg128x160, EPI>(p, s);
      else return -7;
  }
  return -7;
}

unsigned magic_of(int d) { return d > 0 ? (unsigned)((1ull << 32) / (unsigned)d + 1) : 0u; }

}  // namespace

int big_tile_dims(int cfg, int* bm, int* bn) {
  switch (cfg) {
    case BIG_288x256: *bm = 288; *bn = 256; return 0;
    case BIG_256x256: *bm = 256; *bn = 256; return 0;
    case BIG_256x160: *bm = 256; *bn = 160; return 0;
    case BIG_160x256: *bm = 160; *bn = 256; return 0;
    case BIG_256x128: *bm = 256; *bn = 128; return 0;
    case BIG_256x160_K64: *bm = 256; *bn = 160; return 0;
    case BIG_288x256_K64: *bm = 288; *bn = 256; return 0;
    case BIG_256x256_K64: *bm = 256; *bn = 256; return 0;
    case BIG_192x160_K64: *bm = 192; *bn = 160; return 0;
    case BIG_128x160: *bm = 128; *bn = 160; return 0;
    case BIG_256x256_M32: *bm = 256; *bn = 256; return 0;
    case BIG_384x192_M32: *bm = 384; *bn = 192; return 0;
  }
  return -1;
}

int launch_big_nt(int cfg, int epi, const GemmParams& p_in, hipStream_t s) {
  GemmParams p = p_in;
  if (p.K % 32 || p.K < 32 || p.splitk < 1 || p.splitk > 4) return -6;
  if (cfg >= BIG_256x160_K64 && cfg != BIG_128x160 && (p.lda < ((p.K + 63) & ~63) || p.ldb < ((p.K + 63) & ~63) || (p.lda & 7) || (p.ldb & 7))) return -6;
  if ((size_t)p.M * p.lda >= (1ull << 31) || (size_t)p.N * p.ldb >= (1ull << 31)) return -6;  // 32-bit lane offsets
  if (epi == EPI_HEADS) {
    if (p.M >= 65536 || p.N >= 65536) return -8;
    p.ep.mg_hid = magic_of(p.ep.hid);
    p.ep.mg_dh = magic_of(p.ep.dh);
    p.ep.mg_ntok = magic_of(p.ep.n_tok);
  }
  switch (epi) {
    case EPI_BF16: return launch_big_nt_epi<EPI_BF16>(cfg, p, s);
    case EPI_F32_BIAS: return launch_big_nt_epi<EPI_F32_BIAS>(cfg, p, s);
    case EPI_F32_BIAS_RESID: return launch_big_nt_epi<EPI_F32_BIAS_RESID>(cfg, p, s);
    case EPI_BIAS_GELU: return launch_big_nt_epi<EPI_BIAS_GELU>(cfg, p, s);
    case EPI_GELU_BWD: return launch_big_nt_epi<EPI_GELU_BWD>(cfg, p, s);
    case EPI_HEADS: return launch_big_nt_epi<EPI_HEADS>(cfg, p, s);
    case EPI_F32_BF16: return launch_big_nt_epi<EPI_F32_BF16>(cfg, p, s);
  }
  return -7;
}

int launch_skinny_epilogue(int epi, float* acc, int ldacc, const GemmParams& p_in, hipStream_t s) {
  GemmParams p = p_in;
  if (p.ep.ln_h) {  // fused LayerNorm of the output rows (gemm.h EpiParams::ln_*)
    if (epi != EPI_F32_BIAS_RESID || (p.N & 3) || p.N > 1024 || !p.ep.bias || !p.ep.resid || (p.ep.ldo0 & 3) || (p.ep.ldr & 3) ||
        (p.ep.ln_ldh & 3) || !p.ep.ln_g || !p.ep.ln_b || !p.ep.ln_mean || !p.ep.ln_rstd)
      return -9;
    FACT_LAUNCH((skinny_epilogue_ln_kernel<4>), dim3((unsigned)((p.M + 3) / 4)), dim3(256), 0, s, acc, ldacc, p);
    return 0;
  }
  const int total = p.M * ((p.N + 3) >> 2);
  const dim3 grid((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), block(256);
  switch (epi) {
    case EPI_BF16: FACT_LAUNCH(skinny_epilogue_kernel<EPI_BF16>, grid, block, 0, s, acc, ldacc, p); return 0;
    case EPI_F32_BIAS: FACT_LAUNCH(skinny_epilogue_kernel<EPI_F32_BIAS>, grid, block, 0, s, acc, ldacc, p); return 0;
    case EPI_F32_BIAS_RESID:
      FACT_LAUNCH(skinny_epilogue_kernel<EPI_F32_BIAS_RESID>, grid, block, 0, s, acc, ldacc, p); return 0;
    case EPI_BIAS_GELU: FACT_LAUNCH(skinny_epilogue_kernel<EPI_BIAS_GELU>, grid, block, 0, s, acc, ldacc, p); return 0;
    case EPI_GELU_BWD: FACT_LAUNCH(skinny_epilogue_kernel<EPI_GELU_BWD>, grid, block, 0, s, acc, ldacc, p); return 0;
    case EPI_HEADS: FACT_LAUNCH(skinny_epilogue_kernel<EPI_HEADS>, grid, block, 0, s, acc, ldacc, p); return 0;
    case EPI_F32_BF16: FACT_LAUNCH(skinny_epilogue_kernel<EPI_F32_BF16>, grid, block, 0, s, acc, ldacc, p); return 0;
  }
  return -7;
}

// Whole-K grouped wgrad launch.  Every problem: K % 32 == 0, lda / ldb % 8 == 0 and >= 8, out 16-byte
// aligned with ldo % 4 == 0, M % 4 == 0 and N % 4 == 0.  Fills tiles_m / tile_begin.
int g_tn_cfg = 0;  // main loop: 0 = staggered wave groups, 1 = fragment prefetch (block), 2 = prefetch interleaved

template <class C, int PF>
int launch_big_tn_group_t(TnGroup g, hipStream_t s, int parts) {
  if (g.n < 1 || g.n > TN_GROUP_MAX || g.K % 32 || g.K < 32) return -1;
  int total = 0;
  for (int i = 0; i < g.n; ++i) {
    TnProblem& q = g.p[i];
    if ((q.lda & 7) || (q.ldb & 7) || q.lda < 8 || q.ldb < 8 || (q.ldo & 3) || (q.M & 3) || (q.N & 3)) return -2;
    if (((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15) || ((uintptr_t)q.out & 15)) return -5;
    q.tiles_m = (q.M + C::BM - 1) / C::BM;
    q.tile_begin = total;
    total += q.tiles_m * ((q.N + C::BN - 1) / C::BN);
  }
  if (g.adam) {  // fused optimizer epilogue: whole 16-wide units only, a complete set of arenas and shadows per problem
    if (!g.overwrite) return -3;
    for (int i = 0; i < g.n; ++i) {
      const TnProblem& q = g.p[i];
      if ((q.M & 15) || (q.N & 15) || !q.p || !q.m1 || !q.v || !q.sd || !q.st || (q.ldsd & 7) || (q.ldst & 7)) return -3;
      if (((uintptr_t)q.p & 15) || ((uintptr_t)q.m1 & 15) || ((uintptr_t)q.v & 15) || ((uintptr_t)q.sd & 15) || ((uintptr_t)q.st & 15)) return -5;
    }
  }
  static bool once = false;
  if (!once) {
    if (int rc = allow_lds(big_tn_kernel<C, PF, false>, C::LDS_BYTES)) return rc;
    if (int rc = allow_lds(big_tn_kernel<C, PF, true>, C::LDS_BYTES)) return rc;
    once = true;
  }
  if (parts < 1) parts = 1;
  for (int i = 0; i < parts; ++i) {
    const int lo = (int)((long long)total * i / parts), hi = (int)((long long)total * (i + 1) / parts);
    if (hi <= lo) continue;
    g.tile0 = lo;
    if (g.adam) FACT_LAUNCH((big_tn_kernel<C, PF, true>), dim3(hi - lo), dim3(C::NW * 64), C::LDS_BYTES, s, g);
    else FACT_LAUNCH((big_tn_kernel<C, PF, false>), dim3(hi - lo), dim3(C::NW * 64), C::LDS_BYTES, s, g);
  }
  return 0;
}

void gemm_set_tn_cfg(int v) { g_tn_cfg = v; }
bool big_tn_group_has_colsum() { return g_tn_cfg == 0; }

int launch_big_tn_group(TnGroup g, hipStream_t s, int parts) {
  if (g_tn_cfg == 1) return launch_big_tn_group_t<Cfg160x256, 1>(g, s, parts);
  if (g_tn_cfg == 2) return launch_big_tn_group_t<Cfg160x256, 2>(g, s, parts);
  if (g_tn_cfg == 6) return launch_big_tn_group_t<Cfg160x256r6, 2>(g, s, parts);  // interleaved loop, 6-slot ring
  return launch_big_tn_group_t<Cfg160x256, 0>(g, s, parts);
}
