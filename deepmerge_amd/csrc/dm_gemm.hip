// MFMA GEMM family for the DeepMerge pair encoder (gfx950 / CDNA4).
//
// One kernel template covers the three products a Linear layer needs
//   NT  y  = x  W^T   (forward)      A [M,K] k-contiguous,  B [N,K] k-contiguous
//   NN  dx = dy W     (dgrad)        A [M,K] k-contiguous,  B [K,N] n-contiguous
//   TN  dW = dy^T x   (wgrad)        A [K,M] m-contiguous,  B [K,N] n-contiguous
// in two numerics modes: bf16 operands on v_mfma_f32_16x16x32_bf16 (throughput mode) and fp32
// operands on v_mfma_f32_16x16x4_f32 (parity mode; exact fp32 fmaf chains).  Accumulation is
// fp32 in both.
//
// Geometry: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
// 4x4 MFMA tiles of 16x16), K staged through LDS in stages of 128 bytes per row (64 bf16 / 32
// fp32), double-buffered; global->register loads of stage t+1 are issued before the MFMAs of
// stage t and written to LDS after them, one barrier per stage.
//
// LDS images (per operand per stage, 16 KiB + pad):
//   k-contiguous operand  : [128 rows][8 chunks of 16 B], chunk index XOR (row & 7)
//                           -> every ds_read_b128 fragment read is bank-conflict free;
//   m-contiguous operand  : bf16  [64 k-rows][8 slots of 32 B], slot XOR f(k),
//                                 f(k) = (k&3) | ((k>>3)&1)<<2, read with ds_read_b64_tr_b16
//                                 (hardware transpose: 4 k-rows x 16 columns per 16 lanes);
//                           fp32  [32 k-rows][128+4 floats], read with ds_read_b32.
// Fragment convention (both dtypes): a "fragment" is 16 bytes per lane; lane group g = lane>>4
// owns the g-th 16-byte chunk of a 64-byte k-block, i.e. k = 8g..8g+7 (bf16) or 4g..4g+3 (fp32).
// For fp32 the four MFMA k-steps of a block take element j of every lane's chunk, which is a
// permutation of k applied identically to A and B.
//
// The MFMA is issued with the operands swapped (B fragment in the A slot), so a lane ends up
// with 4 CONSECUTIVE columns n of one row m: epilogue loads/stores are 16-byte (fp32) or 8-byte
// (bf16) vectors.
#include <cstdlib>

#include "dm_common.h"
#include <cstring>

#include "dm_gemm_common.h"
#include "dm_mfma.h"
#include "dm_prof.h"

// dm_gemm256.hip: the 256x256 LDS-DMA pipeline for large bf16 products
bool dm_gemm256_plan(GemmParams &p, int layout, int ab_dtype, bool can_split, long long workspace_bytes, int user_split);
void dm_gemm256_launch(const GemmParams &p, int layout, hipStream_t s);
int dm_gemm_ring_plan(GemmParams &p, int layout, int ab_dtype, bool aligned8);      // dm_gemm_ring.hip
void dm_gemm_ring_launch(const GemmParams &p, int wm, hipStream_t s);
int dm_gemm_w4_plan(GemmParams &p, int layout, int ab_dtype, bool aligned8, bool can_split, long long workspace_bytes);   // dm_gemm_w4.hip (grid size, 0 = not taken)
void dm_gemm_w4_launch(const GemmParams &p, int layout, int grid, hipStream_t s);
int dm_gemm_w4_grouped(GemmParams *ps, int n, hipStream_t s, bool launch, const DmGroupedExtra &x);      // dm_gemm_w4.hip: n weight gradients in one launch (0 = not taken, 1 = one K slice per tile, 2 = stream-K)
long long dm_gemm_w4_grouped_ws_bytes();

namespace {

constexpr int BM = 128, BN = 128;
constexpr int STAGE_BYTES = 16896;  // 16 KiB, or 32 rows x 528 B for the padded fp32 m-contiguous image
constexpr int NTHREADS = 256;
// LDS stages per operand.  1 = single LDS buffer + the register staging set as the second stage
// (two barriers per K stage, 33 KiB per workgroup -> 3 workgroups per CU, which makes the tile count of
// the model's GEMMs (768 / 2304 / 3072 tiles at M = 16384) an exact number of rounds); 2 = classic
// double-buffered LDS (one barrier per stage, 66 KiB -> 2 workgroups per CU).
#ifndef DM_GEMM_LDS_STAGES
#define DM_GEMM_LDS_STAGES 1
#endif
constexpr int LDS_STAGES = DM_GEMM_LDS_STAGES;
constexpr int WG_PER_CU = (LDS_STAGES == 1) ? 3 : 2;


// Tile geometry.  TM = 16x16 MFMA tiles per wave per dimension: TM = 4 -> 128x128 workgroup tile, TM = 2 -> 64x64
// (used when a product has too few 128x128 tiles to fill the chip: the M = 4096 / 1024 stages of the encoder).
template <typename T, int TM> struct Geo {
  static constexpr int EPC = DmTypeInfo<T>::kPerChunk;
  static constexpr int TILE = 32 * TM;                         // rows/cols of the workgroup tile
  static constexpr int BK = 128 / (int)sizeof(T);              // k per stage: 64 (bf16) / 32 (fp32)
  // m-contiguous image: one row per k, TILE elements wide
  static constexpr int M_ROWB = (sizeof(T) == 2) ? TILE * 2 : (TILE + 4) * 4;   // bf16: exact; fp32: +4 floats pad
  static constexpr int M_CPR = TILE / EPC;                     // 16-byte chunks per k-row
  static constexpr int M_RPI = NTHREADS / M_CPR;               // k-rows staged per pass
  static constexpr int K_BYTES = TILE * 128;                   // k-contiguous image
  static constexpr int STAGE = (K_BYTES > BK * M_ROWB) ? K_BYTES : BK * M_ROWB;
};

// swizzle of the 32-byte column slots of the bf16 m-contiguous image, chosen so that the 8 rows one half-wave
// touches in a ds_read_b64_tr_b16 (k = q and 8+q, q = 0..3) land in 8 different 32-byte bank groups:
//   256-byte rows (TM = 4): 8 slots per row  -> f(k) = (k&3) | ((k>>3)&1)<<2
//   128-byte rows (TM = 2): 4 slots per row, two rows per 256-byte bank row -> f(k) = ((k>>1)&1) | ((k>>3)&1)<<1
template <int TM> __device__ __forceinline__ int tr_swz(int k) {
  if constexpr (TM == 4) return (k & 3) | (((k >> 3) & 1) << 2);
  else return ((k >> 1) & 1) | (((k >> 3) & 1) << 1);
}

// ---- global -> register staging -------------------------------------------------------------
// Loads go through a buffer descriptor covering exactly the operand's bytes: an out-of-range row
// (tile overhang in M / N, or k >= K for row-per-k operands) reads as ZERO in hardware, so the
// loop carries no branches and no 64-bit address arithmetic -- TM per-thread byte offsets are
// computed once and each K stage only bumps one scalar offset.
template <int TM> struct OperandView {
  __amdgpu_buffer_rsrc_t rsrc;   // k-contiguous operands: descriptor of this tile's row panel
  const void *base;              // m-contiguous operands: tile column origin (row k = 0), re-based every stage
  long long tail_bytes;          // m-contiguous: bytes from `base` to the end of the operand
  unsigned voff[TM];             // per-thread byte offsets of its 16-byte chunks inside the panel / stage
  unsigned chunk_k;              // k index (elements) of this thread's chunk inside a stage (k-contiguous operands)
};

__device__ __forceinline__ int clamp_records(long long bytes) {
  return (int)(bytes < 0 ? 0 : (bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes));
}

// K-contiguous tile: 32*TM rows x 8 chunks; thread t: chunk t&7, rows (t>>3)+32i.  The descriptor starts at the
// tile's first row (64-bit pointer arithmetic once per workgroup), so operands of any size work with 32-bit offsets.
template <typename T, int TM>
__device__ __forceinline__ OperandView<TM> view_kmajor(const T *base, long long ld, int row0, int rows, int K, int t) {
  constexpr int EPC = DmTypeInfo<T>::kPerChunk;
  OperandView<TM> v;
  const int nrows = min(32 * TM, rows - row0);                 // >= 1: the grid never starts a tile past the operand
  const long long bytes = ((long long)(nrows - 1) * ld + K) * (long long)sizeof(T);
  v.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(base + (long long)row0 * ld), 0, clamp_records(bytes), 0x00020000);
  v.base = nullptr;
  v.tail_bytes = 0;
  v.chunk_k = (t & 7) * EPC;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = (t >> 3) + 32 * i;
    const long long off = ((long long)row * ld + v.chunk_k) * (long long)sizeof(T);
    v.voff[i] = (row < nrows) ? (unsigned)off : 0x80000000u;      // past the last row: force out of range
  }
  return v;
}
template <typename T, int TM>
__device__ __forceinline__ void load_kmajor(u32x4 (&r)[TM], const OperandView<TM> &v, int k0, int kend, long long kphys) {
  // chunks at or beyond kend (K tail of the last stage) must read zero, not the next row; kphys = element offset of logical column k0
  // inside a row (k0 itself for a plain operand, segment offset + in-segment column for a folded one)
  const unsigned kill = ((int)(k0 + v.chunk_k) < kend) ? 0u : 0x80000000u;
  const unsigned soff = (unsigned)kphys * (unsigned)sizeof(T);
#pragma unroll
  for (int i = 0; i < TM; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(v.rsrc, v.voff[i] | kill, soff, 0);
}
template <int TM> __device__ __forceinline__ void store_kmajor(char *lds, const u32x4 (&r)[TM], int t) {
  const int c = t & 7;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = (t >> 3) + 32 * i;
    *reinterpret_cast<u32x4 *>(lds + row * 128 + ((c ^ (row & 7)) << 4)) = r[i];
  }
}
// M-contiguous tile: rows are k (64 for bf16, 32 for fp32), 32*TM elements wide.  The descriptor is re-based at
// every stage (scalar work) so that offsets stay small for operands with millions of rows; rows k >= K fall
// outside it and read zero.  Split-K slices end on stage boundaries, so no other k predicate is needed.  Columns
// past the operand's width are killed per thread (they would otherwise alias the next row).
template <typename T, int TM>
__device__ __forceinline__ OperandView<TM> view_mmajor(const T *base, long long ld, int col0, int cols, int K, int t) {
  using G = Geo<T, TM>;
  OperandView<TM> v;
  v.base = base + col0;
  v.tail_bytes = ((long long)(K - 1) * ld + (cols - col0)) * (long long)sizeof(T);
  v.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(base), 0, 0, 0x00020000);
  v.chunk_k = 0;
  const int col = (t % G::M_CPR) * G::EPC;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long k = t / G::M_CPR + G::M_RPI * i;
    v.voff[i] = (col0 + col < cols) ? (unsigned)((k * ld + col) * (long long)sizeof(T)) : 0x80000000u;
  }
  return v;
}
template <typename T, int TM>
__device__ __forceinline__ void load_mmajor(u32x4 (&r)[TM], const OperandView<TM> &v, long long ld, int k0, long long seg_off = 0) {
  // k0 = first row of the stage inside its K segment (the operand itself when not folded), seg_off = the segment's element offset
  const long long skip = (long long)k0 * ld * (long long)sizeof(T);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(v.base) + skip + seg_off * (long long)sizeof(T)), 0, clamp_records(v.tail_bytes - skip), 0x00020000);
#pragma unroll
  for (int i = 0; i < TM; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, v.voff[i], 0, 0);
}
template <typename T, int TM> __device__ __forceinline__ void store_mmajor(char *lds, const u32x4 (&r)[TM], int t) {
  using G = Geo<T, TM>;
  const int c = t % G::M_CPR;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int k = t / G::M_CPR + G::M_RPI * i;
    if constexpr (sizeof(T) == 2)   // 16-byte chunk = 8 columns; two chunks per 32-byte slot
      *reinterpret_cast<u32x4 *>(lds + k * G::M_ROWB + (((c >> 1) ^ tr_swz<TM>(k)) << 5) + ((c & 1) << 4)) = r[i];
    else
      *reinterpret_cast<u32x4 *>(lds + k * G::M_ROWB + (c << 4)) = r[i];
  }
}

// ---- LDS -> fragment reads -------------------------------------------------------------------
// K-contiguous: tile row = wave-local row; same code for both dtypes and tile sizes.
__device__ __forceinline__ u32x4 frag_kmajor(const char *lds, int row, int kb, int lane) {
  const int chunk = kb * 4 + (lane >> 4);
  return *reinterpret_cast<const u32x4 *>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
}
template <typename T, int TM> __device__ __forceinline__ u32x4 frag_mmajor(const char *lds, int col0, int kb, int lane) {
  using G = Geo<T, TM>;
  const int g = lane >> 4;
  u32x4 out;
  if constexpr (sizeof(T) == 2) {
    // ds_read_b64_tr_b16: within a 16-lane group, lane 4q+p supplies the address of row q,
    // columns 4p..4p+3; lane i receives column i of the 4 rows.
    const int q = (lane >> 2) & 3, p = lane & 3;
    const int col = col0 + 4 * p;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int k = kb * 32 + 8 * g + 4 * half + q;
      const u32x2 w = dm_ds_read_tr16(lds + k * G::M_ROWB + (((col >> 4) ^ tr_swz<TM>(k)) << 5) + ((col & 15) << 1));
      out[2 * half] = w[0];
      out[2 * half + 1] = w[1];
    }
  } else {
    const int i = lane & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kb * 16 + 4 * g + j;
      out[j] = *reinterpret_cast<const unsigned int *>(lds + k * G::M_ROWB + ((col0 + i) << 2));
    }
  }
  return out;
}

// FOLD: the operands are hi / lo plane pairs and the contraction runs over three segments of p.k_fold (GemmParams.k_fold; a multiple
// of the K stage, so a stage never straddles two segments and there is no K tail).
template <typename T, int LAYOUT, int TM, bool FOLD = false>
__global__ __launch_bounds__(NTHREADS, (TM == 4 ? WG_PER_CU : 4)) void gemm_kernel(const GemmParams p) {
  using G = Geo<T, TM>;
  constexpr bool A_MMAJOR = (LAYOUT == DM_TN);
  constexpr bool B_MMAJOR = (LAYOUT != DM_NT);
  constexpr int BK = G::BK;
  constexpr int TILE = G::TILE, WT = 16 * TM;     // workgroup tile, wave tile
  __shared__ __attribute__((aligned(16))) char smem[2 * LDS_STAGES * G::STAGE];
  auto ldsA = [&](int buf) -> char * { return smem + (2 * (buf % LDS_STAGES)) * G::STAGE; };
  auto ldsB = [&](int buf) -> char * { return smem + (2 * (buf % LDS_STAGES) + 1) * G::STAGE; };

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = gridDim.x;
  int id = dm_xcd_remap(blockIdx.x, nwg);
  const int per_z = p.tiles_m * p.tiles_n;
  const int z = id / per_z;
  id -= z * per_z;
  int tm, tn;
  if (p.group_m > 0) {
    const int band = id / (p.group_m * p.tiles_n);
    const int within = id - band * (p.group_m * p.tiles_n);
    const int gsz = min(p.group_m, p.tiles_m - band * p.group_m);
    tn = within / gsz;
    tm = band * p.group_m + (within - tn * gsz);
  } else {
    tn = id % p.tiles_n;
    tm = id / p.tiles_n;
  }
  const int m0 = tm * TILE, n0 = tn * TILE;
  const int kbeg = z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);

  const T *A = reinterpret_cast<const T *>(p.A);
  const T *B = reinterpret_cast<const T *>(p.B);

  f32x4 acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Weight gradients on 64 x 64 tiles (the 4096- / 1024-token stages, the patch embeds): the bias gradient = column sums of A = dy rides
  // along as in the 256 x 256 / 4-wave kernels -- the waves wn == 0 of the workgroups of column tile 0 multiply their A fragments with a
  // ones fragment (+50 % MFMAs in those workgroups only); partial rows [z][M] in p.colsum_slab, folded by the split-K reduction.
  // Round 4: nine colsum + nine partial-reduce launches per step gone.
  constexpr bool CS = A_MMAJOR && TM == 2 && sizeof(T) == 2;
  const bool colsum = CS && p.colsum_slab != nullptr && tn == 0 && wn == 0;
  f32x4 accb[CS ? TM : 1];
#pragma unroll
  for (int i = 0; i < (CS ? TM : 1); ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};     // eight bf16 1.0

  u32x4 ra[TM], rb[TM];
  OperandView<TM> va, vb;
  // (folded: a k-contiguous view spans up to the farthest segment's last column, an m-contiguous one the rows of ONE segment)
  const long long a_far = FOLD ? max(p.a_fold[0], max(p.a_fold[1], p.a_fold[2])) : 0, b_far = FOLD ? max(p.b_fold[0], max(p.b_fold[1], p.b_fold[2])) : 0;
  const int k_seg = FOLD ? p.k_fold : p.K;
  if constexpr (A_MMAJOR) va = view_mmajor<T, TM>(A, p.lda, m0, p.M, k_seg, t);
  else va = view_kmajor<T, TM>(A, p.lda, m0, p.M, (int)min((long long)0x7fffffff, a_far + k_seg), t);
  if constexpr (B_MMAJOR) vb = view_mmajor<T, TM>(B, p.ldb, n0, p.N, k_seg, t);
  else vb = view_kmajor<T, TM>(B, p.ldb, n0, p.N, (int)min((long long)0x7fffffff, b_far + k_seg), t);
  auto gload = [&](int k0) {
    int kk = k0;
    long long oa = 0, ob = 0;
    if constexpr (FOLD) {
      const int seg = (k0 >= 2 * p.k_fold) ? 2 : (k0 >= p.k_fold) ? 1 : 0;
      kk = k0 - seg * p.k_fold;
      oa = seg == 0 ? p.a_fold[0] : seg == 1 ? p.a_fold[1] : p.a_fold[2];
      ob = seg == 0 ? p.b_fold[0] : seg == 1 ? p.b_fold[1] : p.b_fold[2];
    }
    if constexpr (A_MMAJOR) load_mmajor<T, TM>(ra, va, p.lda, kk, oa);
    else load_kmajor<T, TM>(ra, va, k0, kend, oa + kk);
    if constexpr (B_MMAJOR) load_mmajor<T, TM>(rb, vb, p.ldb, kk, ob);
    else load_kmajor<T, TM>(rb, vb, k0, kend, ob + kk);
  };
  auto lstore = [&](int buf) {
    if constexpr (A_MMAJOR) store_mmajor<T, TM>(ldsA(buf), ra, t);
    else store_kmajor<TM>(ldsA(buf), ra, t);
    if constexpr (B_MMAJOR) store_mmajor<T, TM>(ldsB(buf), rb, t);
    else store_kmajor<TM>(ldsB(buf), rb, t);
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    gload(kbeg);
    lstore(0);
  }
  __syncthreads();
  // Epilogue read operands (the fp32 residual; the aux of the multiply / GELU' epilogues) are touched towards L2 a few K stages before
  // the tile ends: the whole-line epilogue walks its 64 x 64 block 16 rows at a time and would otherwise pay a first-touch round
  // trip per pass (proj forward reads 50 MB of residual stream, the dgrad of fc2 100 MB of GELU').  One row per lane, one dword
  // per 128-byte line; the values are not read before the epilogue (the asm at its top is the use that keeps the loads alive).
  float tv0 = 0.f, tv1 = 0.f;
  const int touch_at = (TM == 4 && sizeof(T) == 2 && p.split_k <= 1 && !(p.debug & 0x200) && (p.residual || (p.aux && (p.epilogue == DM_EPI_DGELU || p.epilogue == DM_EPI_MUL))))
                           ? max(0, nk - 4) : -1;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1 < nk);
    if (more) gload(kbeg + (kt + 1) * BK);
    if (kt == touch_at) {
      const int m = m0 + wm * WT + lane, n = n0 + wn * WT;
      if (m < p.M && n < p.N) {
        const DmGemmRow rw = dm_gemm_row(p, m);
        if (p.residual) {
          tv0 = p.residual[rw.r + n];
          if (n + 32 < p.N) tv1 = p.residual[rw.r + n + 32];
        } else if (p.aux_dtype == DM_F32) {
          tv0 = reinterpret_cast<const float *>(p.aux)[rw.x + n];
          if (n + 32 < p.N) tv1 = reinterpret_cast<const float *>(p.aux)[rw.x + n + 32];
        } else {
          tv0 = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short *>(p.aux)[rw.x + n]);
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      u32x4 fa[TM], fb[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (A_MMAJOR) fa[i] = frag_mmajor<T, TM>(ldsA(cur), wm * WT + i * 16, kb, lane);
        else fa[i] = frag_kmajor(ldsA(cur), wm * WT + i * 16 + (lane & 15), kb, lane);
        if constexpr (B_MMAJOR) fb[i] = frag_mmajor<T, TM>(ldsB(cur), wn * WT + i * 16, kb, lane);
        else fb[i] = frag_kmajor(ldsB(cur), wn * WT + i * 16 + (lane & 15), kb, lane);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) mma<T>(acc[i][j], fa[i], fb[j]);
      if constexpr (CS) {
        if (colsum && (!FOLD || dm_fold_counts(p, kbeg + kt * BK))) {
#pragma unroll
          for (int i = 0; i < TM; ++i) mma<T>(accb[i], fa[i], ones);
        }
      }
    }
    if constexpr (LDS_STAGES == 1) __syncthreads();   // every wave is done reading the single buffer
    if (more) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----------------------------------------------------------------------------
  const int g = lane >> 4, li = lane & 15;
  if constexpr (CS) {
    if (colsum && g == 0) {        // every column of an accb tile holds the same sum: lanes g == 0 write element 0
      float *row = p.colsum_slab + (long long)z * p.M;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WT + i * 16 + li;
        if (m < p.M) row[m] = accb[i][0];
      }
    }
  }
#ifdef DM_GEMM_ABLATE
  if (p.debug & 0x800) {                 // (ablation builds, DM_GEMM_NOEPI=1: no epilogue at all -- what the K loops alone cost; the asm keeps the MFMAs alive)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
#endif
  if (p.split_k > 1) {
    float *W = p.workspace + (long long)z * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WT + i * 16 + li;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int n = n0 + wn * WT + j * 16 + 4 * g;
        if (n < p.N) dm_store4(W + (long long)m * p.N + n, acc[i][j]);
      }
    }
    return;
  }
  if constexpr (TM == 4 && sizeof(T) == 2) {
    // whole-line epilogue (dm_gemm_common.h, shared with the LDS-DMA kernels): every wave transposes its 64 x 64 block 16 rows at a time
    // through a private piece of the (now idle) operand stage, so bias / residual / aux reads and the C stores are 128-byte rows instead
    // of 8-byte strips of 16 different rows (the dgrad of fc2 reads 100 MB of GELU' that way): in the step 137 -> 120 us for that
    // product, 53 -> 47 us for the proj forward.  (The 64 x 64 variant gained nothing from the same change and keeps its strips.)
    const bool rows_ok = !(p.debug & 0x100) && (p.N % 8 == 0) && (p.ldc % 8 == 0) && (p.aux == nullptr || p.ldaux % 8 == 0) &&
                         (p.rows_per_group == 0 || p.group_stride % 8 == 0) && (p.residual == nullptr || p.ldr % 8 == 0);
    if (rows_ok) {
      static_assert(2 * LDS_STAGES * G::STAGE >= 4 * 16 * DM_EPI_PITCH, "epilogue staging must fit the operand stage");
      asm volatile("" ::"v"(tv0), "v"(tv1));
      dm_epilogue_rows<4, 16>(p, acc, smem + wave * (16 * DM_EPI_PITCH), m0 + wm * WT, n0 + wn * WT, lane);
      return;
    }
  }
#ifndef DM_TM2_STRIPS_OLD      // (A/B builds: the one-step strips below keep the 64 x 64 kernel at 64-76 registers instead of 94; step A/B: no difference)
  if (TM == 2 && !(p.debug & 0x400)) {
    // 64 x 64 tiles (the 4096- / 1024-token stages, the patch embeds): all loads of the wave's four strips first, then the stores
    // (dm_gemm_common.h: a load behind a store waits for the store's acknowledgement)
    DmGemmRow rbs[TM];
    DmStripPre pre[TM][TM];
    bool ok[TM][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WT + i * 16 + li;
      rbs[i] = dm_gemm_row(p, m < p.M ? m : p.M - 1);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int n = n0 + wn * WT + j * 16 + 4 * g;
        ok[i][j] = m < p.M && n < p.N;      // N % 4 == 0 is enforced by the launcher
        if (ok[i][j]) dm_gemm_strip_load(p, rbs[i], n, pre[i][j]);
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
        if (ok[i][j]) dm_gemm_strip_store<sizeof(T) == 2>(p, acc[i][j], rbs[i], n0 + wn * WT + j * 16 + 4 * g, pre[i][j]);
    return;
  }
#endif
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WT + i * 16 + li;
    if (m >= p.M) continue;
    const DmGemmRow rb = dm_gemm_row(p, m);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int n = n0 + wn * WT + j * 16 + 4 * g;
      if (n >= p.N) continue;   // N % 4 == 0 is enforced by the launcher
      dm_gemm_emit<sizeof(T) == 2>(p, acc[i][j], rb, n);
    }
  }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slab[s][i], slices summed in index order.  Workgroups past `main_blocks` fold the
// pipeline's partial column sums of A (cs_rows [cs_R][cs_M] -> cs_out, rows in index order) in the same launch: the bias
// gradient that rides on a weight gradient used to cost a launch of its own (19 per step).
__global__ void splitk_reduce_kernel(const float *__restrict__ slab, float *__restrict__ out, long long ldc,
                                     int M, int N, int S, int accumulate, int main_blocks, const float *__restrict__ cs_rows,
                                     float *__restrict__ cs_out, int cs_M, int cs_R, int cs_accumulate) {
  if ((int)blockIdx.x >= main_blocks) {
    const int m = ((int)blockIdx.x - main_blocks) * blockDim.x + threadIdx.x;
    if (m >= cs_M) return;
    float s0 = cs_accumulate ? cs_out[m] : 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = 0;
    for (; r + 3 < cs_R; r += 4) {
      s0 += cs_rows[(long long)r * cs_M + m];
      s1 += cs_rows[(long long)(r + 1) * cs_M + m];
      s2 += cs_rows[(long long)(r + 2) * cs_M + m];
      s3 += cs_rows[(long long)(r + 3) * cs_M + m];
    }
    for (; r < cs_R; ++r) s0 += cs_rows[(long long)r * cs_M + m];
    cs_out[m] = (s0 + s1) + (s2 + s3);
    return;
  }
  const long long n4 = (long long)M * N / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)main_blocks * blockDim.x) {
    const long long e = i * 4;
    const int m = (int)(e / N), n = (int)(e % N);
    float *o = out + (long long)m * ldc + n;
    // slices summed in index order, but LOADED eight (then four) at a time: with one load per trip of a run-time loop a thread pays S
    // dependent memory round trips for its single output group -- the kernel was latency-bound at ~6 x 1.3 us, not bandwidth-bound
    // (slabs are read exactly once, here: non-temporal)
    auto ntl = [](const float *q) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(q)); };
    const float *sl = slab + e;
    const long long mn = (long long)M * N;
    f32x4 v = accumulate ? ntl(o) : (f32x4){0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = ntl(sl + (long long)(s + u) * mn);
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    if (s + 4 <= S) {
      f32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = ntl(sl + (long long)(s + u) * mn);
#pragma unroll
      for (int u = 0; u < 4; ++u) v += t[u];
      s += 4;
    }
    if (s + 2 <= S) {
      const f32x4 t0 = ntl(sl + (long long)s * mn), t1 = ntl(sl + (long long)(s + 1) * mn);
      v += t0;
      v += t1;
      s += 2;
    }
    if (s < S) v += ntl(sl + (long long)s * mn);
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(o));      // (the gradient is next read by Adam, a whole backward pass later)
  }
}


// Forward / dgrad K slices: 8 consecutive outputs of a row = sum of the S partial tiles (slice order) through the fused epilogue
// (bias, GELU / GELU' with the saved operand, residual, bf16 / fp32 store, grouped rows): same arithmetic as an unsplit launch except
// for the order of the fp32 additions over K.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmParams p, const float *__restrict__ slab, int S) {
  const long long n8 = (long long)p.M * p.N / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 8;
    const int m = (int)(e / p.N), n = (int)(e % p.N);
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < S; ++sl) {
      lo += dm_load4(slab + (long long)sl * p.M * p.N + e);
      hi += dm_load4(slab + (long long)sl * p.M * p.N + e + 4);
    }
    dm_gemm_emit8(p, lo, hi, dm_gemm_row(p, m), n);
  }
}

// out[m] (+)= sum_r rows[r][m], rows in index order (the pipeline's partial column sums of A)
__global__ void colsum_rows_reduce_kernel(const float *__restrict__ rows, float *__restrict__ out, int M, int R, int accumulate) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float s0 = accumulate ? out[m] : 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;     // independent accumulators: loads overlap
  int r = 0;
  for (; r + 3 < R; r += 4) {
    s0 += rows[(long long)r * M + m];
    s1 += rows[(long long)(r + 1) * M + m];
    s2 += rows[(long long)(r + 2) * M + m];
    s3 += rows[(long long)(r + 3) * M + m];
  }
  for (; r < R; ++r) s0 += rows[(long long)r * M + m];
  out[m] = (s0 + s1) + (s2 + s3);
}

// Generic fp32 kernel for the small / odd-shaped products of the tail (K = 19, N = 100, M = batch): any
// shape, any alignment.  These products have few output tiles and a long K, so they are latency-bound:
// the four waves of a workgroup split K four ways for one 16x16 output tile (each wave stages its own
// 16x32 / 32x16 operand slices through a private LDS region), and the partial tiles are summed in a
// fixed order before the epilogue.  FMA chains in k order within a wave; deterministic.
// NW waves split K; the partial tiles are summed in wave order: deterministic for a given NW.
// one output of the generic fp32 path through the fused epilogue
__device__ __forceinline__ void sgemm_small_emit(const GemmParams &p, int m, int n, float v) {
  if (p.bias) v += p.bias[n];
  const long long ro = (long long)m;
  if (p.epilogue == DM_EPI_GELU) {
    if (p.aux) reinterpret_cast<float *>(p.aux)[ro * p.ldaux + n] = v;
    v = dm_gelu(v);
  } else if (p.epilogue == DM_EPI_DGELU) {
    v *= dm_dgelu(reinterpret_cast<const float *>(p.aux)[ro * p.ldaux + n]);
  } else if (p.epilogue == DM_EPI_GELU_GRAD) {
    reinterpret_cast<float *>(p.aux)[ro * p.ldaux + n] = dm_dgelu(v);
    v = dm_gelu(v);
  } else if (p.epilogue == DM_EPI_MUL) {
    v *= reinterpret_cast<const float *>(p.aux)[ro * p.ldaux + n];
  }
  if (p.residual) v += p.residual[ro * p.ldr + n];
  float *c = reinterpret_cast<float *>(p.C) + ro * p.ldc + n;
  if (p.accumulate) v += *c;
  *c = v;
}

template <int LAYOUT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void sgemm_small_kernel(const GemmParams p) {
  __shared__ float sa[NW][32][17], sb[NW][32][17];
  __shared__ float red[NW][16][17];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
  const float *A = reinterpret_cast<const float *>(p.A);
  const float *B = reinterpret_cast<const float *>(p.B);
  // gridDim.z > 1 (skinny products with a long contraction -- the 100-wide head over 3840 features: 28 tiles): the contraction is cut
  // into gridDim.z slices of p.k_per_split, partial tiles go to p.workspace [z][M][N], sgemm_small_reduce_kernel sums them in slice
  // order and applies the epilogue
  const int kz0 = gridDim.z > 1 ? blockIdx.z * p.k_per_split : 0, kz1 = gridDim.z > 1 ? min(p.K, kz0 + p.k_per_split) : p.K;
  const int kq = ((kz1 - kz0 + NW - 1) / NW + 31) / 32 * 32;    // K slice per wave, multiple of 32
  const int kbeg = kz0 + w * kq, kend = min(kz1, kbeg + kq);
  const int tx = lane & 15, ty = lane >> 4;                // lane computes rows 4ty..4ty+3 of column tx
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float ra[8], rb[8];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = lane + 64 * i;
      {   // A(op)[m0 + r][k0 + kk]: consecutive lanes walk the contiguous index of the layout
        const int r = (LAYOUT == DM_TN) ? (idx & 15) : (idx >> 5), kk = (LAYOUT == DM_TN) ? (idx >> 4) : (idx & 31);
        const int m = m0 + r, k = k0 + kk;
        float v = 0.f;
        if (m < p.M && k < kend) v = (LAYOUT == DM_TN) ? A[(long long)k * p.lda + m] : A[(long long)m * p.lda + k];
        ra[i] = v;
      }
      {   // B(op)[k0 + kk][n0 + c]
        const int c = (LAYOUT == DM_NT) ? (idx >> 5) : (idx & 15), kk = (LAYOUT == DM_NT) ? (idx & 31) : (idx >> 4);
        const int n = n0 + c, k = k0 + kk;
        float v = 0.f;
        if (n < p.N && k < kend) v = (LAYOUT == DM_NT) ? B[(long long)n * p.ldb + k] : B[(long long)k * p.ldb + n];
        rb[i] = v;
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = lane + 64 * i;
      const int ar = (LAYOUT == DM_TN) ? (idx & 15) : (idx >> 5), ak = (LAYOUT == DM_TN) ? (idx >> 4) : (idx & 31);
      sa[w][ak][ar] = ra[i];
      const int bc = (LAYOUT == DM_NT) ? (idx >> 5) : (idx & 15), bk = (LAYOUT == DM_NT) ? (idx & 31) : (idx >> 4);
      sb[w][bk][bc] = rb[i];
    }
  };
  const int nk = (kq + 31) / 32;                             // same trip count for every wave (barriers inside)
  gload(kbeg);
  for (int it = 0; it < nk; ++it) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (it + 1 < nk) gload(kbeg + (it + 1) * 32);
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      const float bv = sb[w][kk][tx];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(sa[w][kk][4 * ty + j], bv, acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[w][4 * ty + j][tx] = acc[j];
  __syncthreads();
  if (threadIdx.x >= 256) return;                              // 16 x 16 outputs
  const int om = threadIdx.x >> 4, on = threadIdx.x & 15;
  const int m = m0 + om, n = n0 + on;
  if (m >= p.M || n >= p.N) return;
  float v = (red[0][om][on] + red[1][om][on]) + (red[2][om][on] + red[3][om][on]);
  if constexpr (NW > 4) {
#pragma unroll
    for (int q = 4; q < NW; q += 4) v += (red[q][om][on] + red[q + 1][om][on]) + (red[q + 2][om][on] + red[q + 3][om][on]);
  }
  if (gridDim.z > 1) {
    p.workspace[((long long)blockIdx.z * p.M + m) * p.N + n] = v;
    return;
  }
  sgemm_small_emit(p, m, n, v);
}

// sum of the K slices of a skinny product (slice order: deterministic) + the fused epilogue; one output per thread
__global__ __launch_bounds__(256) void sgemm_small_reduce_kernel(const GemmParams p, int slices) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x, mn = (long long)p.M * p.N;
  if (i >= mn) return;
  float v = p.workspace[i];
  int z = 1;
  for (; z + 8 <= slices; z += 8) {        // eight slices in flight (a one-by-one loop pays a memory round trip per slice); same order of sums
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = p.workspace[(z + u) * mn + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; z < slices; ++z) v += p.workspace[z * mn + i];
  sgemm_small_emit(p, (int)(i / p.N), (int)(i % p.N), v);
}

template <typename T, int TM>
void launch_mfma(const GemmParams &p, int layout, int grid, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    if (p.k_fold > 0) {
      switch (layout) {
        case DM_NT: hipLaunchKernelGGL((gemm_kernel<T, DM_NT, TM, true>), dim3(grid), dim3(NTHREADS), 0, s, p); break;
        case DM_NN: hipLaunchKernelGGL((gemm_kernel<T, DM_NN, TM, true>), dim3(grid), dim3(NTHREADS), 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_kernel<T, DM_TN, TM, true>), dim3(grid), dim3(NTHREADS), 0, s, p); break;
      }
      return;
    }
  }
  switch (layout) {
    case DM_NT: hipLaunchKernelGGL((gemm_kernel<T, DM_NT, TM>), dim3(grid), dim3(NTHREADS), 0, s, p); break;
    case DM_NN: hipLaunchKernelGGL((gemm_kernel<T, DM_NN, TM>), dim3(grid), dim3(NTHREADS), 0, s, p); break;
    default: hipLaunchKernelGGL((gemm_kernel<T, DM_TN, TM>), dim3(grid), dim3(NTHREADS), 0, s, p); break;
  }
}

// DM_GEMM_ROUTE (see dm_gemm): per-product family override for in-step A/B runs; restores the environment when it goes out of scope.
struct DmRouteOverride {
  static constexpr int NKEY = 5;
  const char *keys[NKEY] = {"DM_GEMM_W4", "DM_GEMM_W4_TN", "DM_GEMM_RING", "DM_GEMM_256", "DM_GEMM_FORCE_TILE"};
  char saved[NKEY][16];
  bool had[NKEY];
  bool active = false;
  DmRouteOverride(int layout, int M, int N, int K) {
    static const char *const e = getenv("DM_GEMM_ROUTE");      // read once: unset (the product's case) costs nothing per call
    if (!e || !*e) return;
    char want[64];
    snprintf(want, sizeof(want), "%s:%dx%dx%d=", layout == DM_NT ? "NT" : layout == DM_NN ? "NN" : "TN", M, N, K);
    const char *hit = strstr(e, want);
    if (!hit) return;
    const char *fam = hit + strlen(want);
    const char *vals[NKEY] = {"0", "0", "0", "0", nullptr};
    if (!strncmp(fam, "w4", 2)) { vals[0] = "2"; vals[1] = "2"; }
    else if (!strncmp(fam, "ring", 4)) vals[2] = "2";
    else if (!strncmp(fam, "256", 3)) vals[3] = "2";
    else if (!strncmp(fam, "128", 3)) vals[4] = "128";
    else if (!strncmp(fam, "64", 2)) vals[4] = "64";
    else return;
    active = true;
    for (int i = 0; i < NKEY; ++i) {
      const char *old = getenv(keys[i]);
      had[i] = old != nullptr;
      snprintf(saved[i], sizeof(saved[i]), "%s", old ? old : "");
      if (vals[i]) setenv(keys[i], vals[i], 1); else unsetenv(keys[i]);
    }
  }
  ~DmRouteOverride() {
    if (!active) return;
    for (int i = 0; i < NKEY; ++i) {
      if (had[i]) setenv(keys[i], saved[i], 1); else unsetenv(keys[i]);
    }
  }
};

// 64x64 tiles when the product has too few 128x128 tiles to give every CU its share
// Measured on MI355X (tools/microbench.py, A/B in one process): 64x64 wins only when there are fewer 128x128 tiles than
// CUs; for wgrad (long contraction, small output) 128x128 + split-K stays ahead unless the contraction is short.
inline int pick_tile(int layout, int M, int N, int K) {
  if (const char *f = getenv("DM_GEMM_FORCE_TILE")) {   // tuning / A-B aid: 64 or 128
    const int v = atoi(f);
    if (v == 64 || v == 128) return v;
  }
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  if (t128 >= 256) return 128;
  // (round 5, tools/routing_check.py: the 768 x 768 weight gradient -- 36 tiles of 128x128 -- at K = 5120 .. 12288 runs 23-36 us on 64x64
  // tiles + split-K against 34-43 us on 128x128; the wider gradients go to the 4-wave kernel before this rule is asked)
  return (layout == DM_TN && K > 4096 && t128 >= 64) ? 128 : 64;
}

int choose_split(int tiles, int K, int bk) {
  // Fill ~2 workgroups per CU on 256 CUs, keep >= 8 stages per slice.
  int s = 1;
  while (tiles * s < 256 * WG_PER_CU && s < 32 && K / (s * 2) >= 8 * bk) s *= 2;
  return s;
}

// Forward / dgrad products with a long contraction and a small output (the 4096- and 1024-token stages' fc2 forward, fc1 / qkv
// dgrad, the 256-pixel patch embed: M x N = 4096 x 768 or 1024 x 768 with K = 2304 .. 4096) have too few 128 x 128 tiles for the
// chip and, as 64 x 64 tiles, move twice the operand bytes through L2 -> LDS (measured 403-440 TFLOP/s, 120-150 at M = 1024, where
// 192 workgroups walk 48 K stages each).  Round 4: K slices for them too -- fp32 partial tiles into the workspace, summed in slice
// order and passed through the fused epilogue by splitk_epilogue_kernel.  Returns the slice count (1 = no split) and the tile.
inline int plan_fwd_split(int layout, int M, int N, int K, int &tile) {
  tile = 0;
  static const int mode = [] { const char *e = getenv("DM_GEMM_FWD_SPLIT"); return e ? atoi(e) : 1; }();      // 0 = off (A/B runs)
  if (mode == 0 || layout == DM_TN || N % 8 != 0 || K < 1536) return 1;
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  if (t128 >= 256) return 1;                       // enough tiles without slices
  int split = 4;
  while (split > 1 && K / split < 8 * 64) split >>= 1;      // >= 8 K stages per slice
  if (split <= 1) return 1;
  // (128 x 128 tiles x 4 slices for the M = 4096 products measured no faster than unsplit 64 x 64 tiles -- the slab round trip eats the
  // gain; those go to the 4-wave kernel's slices, dm_gemm_w4_plan -- so this path serves what is left: M <= 1024)
  if (t128 * split >= 512) return 1;
  tile = 64;
  return split;
}

}  // namespace

// floats at the end of the workspace reserved for the column sums of A (colsum_a): the pipeline's [32 slices * 4][M] partial
// rows, or the standalone kernels' scratch
static int64_t colsum_region_floats(int M) {
  const int64_t fused = 128LL * M, alone = dm_colsum_partial_floats(M);
  return (fused > alone ? fused : alone) + 64;
}

extern "C" int64_t dm_gemm_workspace_bytes(int32_t layout, int32_t M, int32_t N, int32_t K) {
  if (layout != DM_TN) {
    // forward / dgrad K slices (plan_fwd_split, dm_gemm_w4_plan): at most 4 partial tiles
    const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
    const long long t16 = (long long)((M + 15) / 16) * ((N + 15) / 16);
    const int64_t skinny = (K >= 1024 && t16 <= 128) ? (int64_t)32 * M * N * 4 : 0;      // K slices of the generic fp32 path
    const int64_t sliced = (N % 8 == 0 && K >= 1536 && t128 < 256) ? (int64_t)4 * M * N * 4 : 0;
    return skinny > sliced ? skinny : sliced;
  }
  const int tile = pick_tile(layout, M, N, K);
  const int tiles = ((M + tile - 1) / tile) * ((N + tile - 1) / tile);
  const int s = choose_split(tiles, K, 32);
  return (s > 1 ? (int64_t)s * M * N * 4 : 0) + colsum_region_floats(M) * 4;
}

extern "C" int dm_gemm(const DmGemmArgs *a, void *stream) {
  DM_REQUIRE(a != nullptr, DM_ERR_BAD_SHAPE, "dm_gemm: null args");
  DM_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, DM_ERR_BAD_SHAPE, "dm_gemm: M,N,K must be positive (got %d,%d,%d)", a->M, a->N, a->K);
  DM_REQUIRE(a->layout >= DM_NT && a->layout <= DM_TN, DM_ERR_BAD_SHAPE, "dm_gemm: bad layout %d", a->layout);
  DM_REQUIRE(a->ab_dtype == DM_F32 || a->ab_dtype == DM_BF16, DM_ERR_BAD_DTYPE, "dm_gemm: bad ab_dtype %d", a->ab_dtype);
  DM_REQUIRE(a->c_dtype == DM_F32 || a->c_dtype == DM_BF16 || a->c_dtype == DM_BF16_PAIR, DM_ERR_BAD_DTYPE, "dm_gemm: bad c_dtype %d", a->c_dtype);
  if (a->c_dtype == DM_BF16_PAIR)
    DM_REQUIRE(a->layout != DM_TN && !a->accumulate && a->split_k <= 1 && a->c_plane > 0 && a->c_plane % 8 == 0 && a->ldc % 8 == 0 && a->N % 8 == 0 &&
                   a->ab_dtype == DM_BF16 && a->rows_per_group == 0,
               DM_ERR_UNSUPPORTED, "dm_gemm: a plane-pair result needs an NT / NN bf16 product without accumulate, N, ldc and c_plane multiples of 8");
  DM_REQUIRE(a->A && a->B && a->C, DM_ERR_BAD_SHAPE, "dm_gemm: null operand");
  DM_REQUIRE(!(a->accumulate && a->c_dtype != DM_F32), DM_ERR_BAD_DTYPE, "dm_gemm: accumulate needs an fp32 C");
  DM_REQUIRE(!(a->accumulate && (a->epilogue == DM_EPI_DGELU || a->epilogue == DM_EPI_MUL)), DM_ERR_UNSUPPORTED,
             "dm_gemm: accumulate cannot be combined with an aux-reading epilogue (DM_EPI_DGELU / DM_EPI_MUL)");
  DM_REQUIRE(a->epilogue == DM_EPI_NONE || a->epilogue == DM_EPI_GELU || a->aux != nullptr, DM_ERR_BAD_SHAPE,
             "dm_gemm: this epilogue needs aux (only DM_EPI_GELU may run without one: inference)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  GemmParams p{};
  p.A = a->A; p.B = a->B; p.C = a->C; p.bias = a->bias; p.residual = a->residual; p.aux = a->aux;
  p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldaux = a->ldaux;
  p.group_stride = a->group_stride; p.rows_per_group = a->rows_per_group;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.epilogue = a->epilogue; p.accumulate = a->accumulate; p.c_dtype = a->c_dtype; p.aux_dtype = a->aux_dtype;

  // folded contraction (hi / lo plane pairs of the "bf16x3" products): three K segments of k_fold, each a plain operand at its offset
  p.c_plane = a->c_dtype == DM_BF16_PAIR ? a->c_plane : 0;
  p.k_fold = 0;
  if (a->k_fold > 0) {
    DM_REQUIRE(a->ab_dtype == DM_BF16 && a->K == 3 * a->k_fold && a->k_fold % 64 == 0, DM_ERR_UNSUPPORTED,
               "dm_gemm: k_fold needs bf16 operands, K == 3 * k_fold and k_fold %% 64 == 0 (K=%d k_fold=%d)", a->K, a->k_fold);
    for (int sgm = 0; sgm < 3; ++sgm) {
      DM_REQUIRE(a->a_fold[sgm] >= 0 && a->a_fold[sgm] < (1LL << 30) && a->b_fold[sgm] >= 0 && a->b_fold[sgm] < (1LL << 30) &&
                     a->a_fold[sgm] % 8 == 0 && a->b_fold[sgm] % 8 == 0,
                 DM_ERR_UNSUPPORTED, "dm_gemm: k_fold segment offsets must be multiples of 8 in [0, 2^30)");
      p.a_fold[sgm] = a->a_fold[sgm];
      p.b_fold[sgm] = a->b_fold[sgm];
    }
    p.k_fold = a->k_fold;
  }
  const bool folded = p.k_fold > 0;
  // the tail of the workspace belongs to the column sums of A, the rest to the split-K slab
  float *cs_region = nullptr;
  int64_t slab_bytes = a->workspace_bytes;
  if (a->colsum_a) {
    DM_REQUIRE(a->layout == DM_TN, DM_ERR_UNSUPPORTED, "dm_gemm: colsum_a goes with DM_TN (column sums of A = dy)");
    const int64_t need = colsum_region_floats(a->M) * 4;
    DM_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= need, DM_ERR_BAD_SHAPE,
               "dm_gemm: colsum_a needs %lld bytes of workspace (got %lld)", (long long)need, (long long)a->workspace_bytes);
    slab_bytes = (a->workspace_bytes - need) & ~15LL;
    cs_region = reinterpret_cast<float *>(reinterpret_cast<char *>(a->workspace) + slab_bytes);
  }
  // Which extents must be chunk (16-byte) multiples for the MFMA path.
  {
    const long long esz = (a->ab_dtype == DM_BF16) ? 2 : 4;
    DM_REQUIRE(128 * a->lda * esz < (1LL << 31) && 128 * a->ldb * esz < (1LL << 31), DM_ERR_BAD_SHAPE,
               "dm_gemm: leading dimensions too large for 32-bit panel offsets (lda=%lld ldb=%lld)", (long long)a->lda, (long long)a->ldb);
  }
  const int epc = (a->ab_dtype == DM_BF16) ? 8 : 4;
  const int a_inner = (a->layout == DM_TN) ? a->M : a->K;   // contiguous extent of A rows
  const int b_inner = (a->layout == DM_NT) ? a->K : a->N;
  const bool mfma_ok = (a_inner % epc == 0) && (b_inner % epc == 0) && (a->lda % epc == 0) && (a->ldb % epc == 0) &&
                       dm_aligned16(a->A) && dm_aligned16(a->B) && (a->N % 4 == 0) && (a->ldc % 4 == 0) &&
                       (a->residual == nullptr || (a->ldr % 4 == 0 && dm_aligned16(a->residual))) &&
                       (a->aux == nullptr || a->ldaux % 4 == 0) && (a->bias == nullptr || dm_aligned16(a->bias)) &&
                       dm_aligned16(a->C) && (a->rows_per_group == 0 || a->group_stride % 4 == 0) &&
                       // skinny fp32 products of the tail (M = batch) have too few 128x128 tiles to fill the chip
                       !(a->ab_dtype == DM_F32 && a->layout != DM_TN && a->rows_per_group == 0 && a->c_dtype == DM_F32 &&
                         (a->aux == nullptr || a->aux_dtype == DM_F32) &&
                         ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN) < 16);
  if (!mfma_ok) {
    DM_REQUIRE(!folded && a->c_dtype != DM_BF16_PAIR, DM_ERR_UNSUPPORTED, "dm_gemm: k_fold / a plane-pair result need the MFMA path's alignment (M=%d N=%d K=%d)", a->M, a->N, a->K);
    DM_REQUIRE(a->ab_dtype == DM_F32 && a->c_dtype == DM_F32 && (a->aux == nullptr || a->aux_dtype == DM_F32),
               DM_ERR_BAD_ALIGN, "dm_gemm: shape/alignment needs the generic path, which is fp32-only "
               "(M=%d N=%d K=%d lda=%lld ldb=%lld)", a->M, a->N, a->K, (long long)a->lda, (long long)a->ldb);
    DM_REQUIRE(a->rows_per_group == 0, DM_ERR_UNSUPPORTED, "dm_gemm: grouped rows need the MFMA path");
    dim3 grid((a->N + 15) / 16, (a->M + 15) / 16);
    // skinny products with a long contraction (the head: 64 x 100 over 3840 features = 28 tiles, 36 us on 28 workgroups): K slices
    // over gridDim.z, summed in slice order by a second small launch.  DM_GEMM_SKINNY=0 for A/B runs.
    static const bool skinny_on = [] { const char *e = getenv("DM_GEMM_SKINNY"); return !(e && atoi(e) == 0); }();
    const long long tiles16 = (long long)grid.x * grid.y;
    int slices = 1;
    if (skinny_on && a->workspace && !a->colsum_a && a->K >= 1024 && tiles16 <= 128) {
      slices = (int)(512 / tiles16);
      if (slices > a->K / 128) slices = a->K / 128;
      if (slices > 32) slices = 32;
      while (slices > 1 && (int64_t)slices * a->M * a->N * 4 > slab_bytes) slices >>= 1;
    }
    if (slices > 1) {
      p.workspace = reinterpret_cast<float *>(a->workspace);
      p.k_per_split = ((a->K + slices - 1) / slices + 31) / 32 * 32;
      slices = (a->K + p.k_per_split - 1) / p.k_per_split;
      grid.z = slices;
    }
    switch (a->layout) {
      // (a 16-wave instance -- K split 16 ways, 8 stages instead of 30 for the 3840-wide head product -- measured 92 us against ~36 us for
      // this one: not used)
      case DM_NT: hipLaunchKernelGGL((sgemm_small_kernel<DM_NT>), grid, dim3(256), 0, s, p); break;
      case DM_NN: hipLaunchKernelGGL((sgemm_small_kernel<DM_NN>), grid, dim3(256), 0, s, p); break;
      default: hipLaunchKernelGGL((sgemm_small_kernel<DM_TN>), grid, dim3(256), 0, s, p); break;
    }
    if (slices > 1)
      hipLaunchKernelGGL(sgemm_small_reduce_kernel, dim3((unsigned)(((long long)a->M * a->N + 255) / 256)), dim3(256), 0, s, p, slices);
    DM_LAUNCH_CHECK("dm_gemm(generic)");
    if (a->colsum_a) return dm_colsum(a->A, a->ab_dtype, a->lda, a->colsum_a, a->K, a->M, a->colsum_accumulate, cs_region, stream);
    return DM_OK;
  }

  const bool can_split = (a->layout == DM_TN) && a->epilogue == DM_EPI_NONE && !a->bias && !a->residual &&
                         a->c_dtype == DM_F32 && a->rows_per_group == 0 && a->workspace != nullptr && slab_bytes > 0;
  if (a->split_k > 1) {
    DM_REQUIRE(can_split, DM_ERR_UNSUPPORTED, "dm_gemm: split_k needs DM_TN, no epilogue, fp32 C and a workspace");
    // a caller-chosen slice count writes split_k fp32 copies of C into the workspace: refuse rather than run past its end
    // (found by tools/mb_wgrad_group.py: a user split above the automatic one faulted on the GPU)
    DM_REQUIRE((int64_t)a->split_k * a->M * a->N * 4 <= slab_bytes, DM_ERR_BAD_SHAPE,
               "dm_gemm: split_k=%d needs %lld bytes of split-K workspace, got %lld", a->split_k,
               (long long)((int64_t)a->split_k * a->M * a->N * 4), (long long)slab_bytes);
  }
  // two-workgroups-per-CU ring kernel (k-contiguous operands, 16-byte row pieces in the epilogue)
  const bool ring_aligned = (a->ldc % 8 == 0) && (a->aux == nullptr || a->ldaux % 8 == 0) &&
                            (a->rows_per_group == 0 || a->group_stride % 8 == 0) && a->split_k <= 1 && !a->colsum_a;
  // the 4-wave register-staged persistent kernel, then the ring kernel, then the 256x256 pipeline
  p.workspace = reinterpret_cast<float *>(a->workspace);
  const bool w4_ok = (a->layout == DM_TN) ? (a->split_k == 0 && a->ldc % 4 == 0) : ring_aligned;     // wgrad: automatic slice count only
  // forward / dgrad K slices (w4: (tile, slice) per workgroup; 128 x 128 / 64 x 64: plan_fwd_split): automatic slice count only, the
  // 8-column epilogue of splitk_epilogue_kernel must be legal, and the caller's workspace holds the slab
  const bool fwd_slices_ok = a->layout != DM_TN && a->ab_dtype == DM_BF16 && a->split_k == 0 && ring_aligned && a->N % 8 == 0 &&
                             (a->residual == nullptr || a->ldr % 8 == 0) && a->workspace != nullptr && slab_bytes > 0;
  // A/B aid (tools/routing_check.py finds candidates in a cold microbenchmark; the decision is taken INSIDE the step): DM_GEMM_ROUTE names a
  // kernel family for single products, e.g. "NT:16384x2304x768=ring,NN:16384x3072x768=256" (families: w4, ring, 256, 128, 64).  The plans
  // below read their switches per call, so the override sets them for this call only.  Not thread-safe; never set in production.
  DmRouteOverride route_guard(a->layout, a->M, a->N, a->K);
  const int w4 = w4_ok ? dm_gemm_w4_plan(p, a->layout, a->ab_dtype, true, a->layout == DM_TN ? can_split : fwd_slices_ok, slab_bytes) : 0;
  const bool persistent = w4 != 0;
  const int ring = persistent ? 0 : dm_gemm_ring_plan(p, a->layout, a->ab_dtype, ring_aligned);
  const bool big = !persistent && !ring && dm_gemm256_plan(p, a->layout, a->ab_dtype, can_split, slab_bytes, a->split_k);
  int tile = w4 ? 1924 : ring ? (ring == 8 ? 2568 : 1288) : big ? 256 : pick_tile(a->layout, a->M, a->N, a->K);
  int split = w4 ? p.split_k : (ring || persistent) ? 1 : p.split_k;
  // forward / dgrad K slices (plan_fwd_split): bf16, automatic slice count, 8-column epilogue legal, slab inside the workspace
  bool fwd_split = persistent && a->layout != DM_TN && p.split_k > 1;      // the 4-wave kernel planned slices
  if (!big && !ring && !persistent && fwd_slices_ok) {
    int ft;
    const int fs = plan_fwd_split(a->layout, a->M, a->N, a->K, ft);
    if (fs > 1 && (int64_t)fs * a->M * a->N * 4 <= slab_bytes && !getenv("DM_GEMM_FORCE_TILE")) {
      fwd_split = true;
      tile = ft;
    }
  }
  if (!big && !ring && !persistent) {
    p.tiles_m = (a->M + tile - 1) / tile;
    p.tiles_n = (a->N + tile - 1) / tile;
    const int bk = (a->ab_dtype == DM_BF16) ? 64 : 32;
    split = a->split_k;
    if (fwd_split) { int ft; split = plan_fwd_split(a->layout, a->M, a->N, a->K, ft); }
    else if (split == 0) split = can_split ? choose_split(p.tiles_m * p.tiles_n, a->K, bk) : 1;
    if (split > 1)
      while (split > 1 && (int64_t)split * a->M * a->N * 4 > slab_bytes) split >>= 1;
    int kps = ((a->K + split - 1) / split + bk - 1) / bk * bk;
    split = (a->K + kps - 1) / kps;
    p.split_k = split;
    p.k_per_split = kps;
  }
  p.workspace = reinterpret_cast<float *>(a->workspace);
  {
    static const bool rows_off = [] { const char *e = getenv("DM_GEMM_T128_ROWS"); return e && e[0] == '0'; }();   // A/B aid: 4-column epilogue in the 128x128 kernel
    if (rows_off && !ring) p.debug |= 0x100;
    static const bool touch_off = [] { const char *e = getenv("DM_GEMM_T128_TOUCH"); return e && e[0] == '0'; }();  // A/B aid: no early touch of the epilogue operands
    if (touch_off) p.debug |= 0x200;
    static const bool lean_off = [] { const char *e = getenv("DM_GEMM_EPI_LEAN"); return e && e[0] == '0'; }();     // A/B aid: the generic whole-line epilogue (dm_gemm_common.h)
    if (lean_off) p.debug |= 0x400;
#ifdef DM_GEMM_ABLATE
    { const char *e = getenv("DM_GEMM_NOEPI"); if (e && e[0] == '1') p.debug |= 0x800; }
#endif
  }
  {
    static const int forced = [] { const char *e = getenv("DM_GEMM_GROUP_M"); return e ? atoi(e) : -1; }();
    p.group_m = forced >= 0 ? forced : 8;   // measured over the encoder's step: 8 > 4 > 0 (column-fastest) > 16 > 32, within 2 %
  }
  const int grid = p.tiles_m * p.tiles_n * split;
  {
    static const char *kNames[2][3] = {{"gemm_f32_NT", "gemm_f32_NN", "gemm_f32_TN"}, {"gemm_bf16_NT", "gemm_bf16_NN", "gemm_bf16_TN"}};
    const double esz = (a->ab_dtype == DM_BF16) ? 2.0 : 4.0;
    const double csz = (a->c_dtype == DM_BF16) ? 2.0 : 4.0;
    // DM_PROF_SHAPES=1: one profiler row per (layout, shape, epilogue) instead of per layout (tuning aid)
    static const bool by_shape = [] { const char *e = getenv("DM_PROF_SHAPES"); return e && e[0] == '1'; }();
    char shaped[64];
    const char *pname = kNames[a->ab_dtype == DM_BF16][a->layout];
    if (by_shape) {
      snprintf(shaped, sizeof(shaped), "%s_%lldx%lldx%lld_e%d%s%s_t%d", pname, (long long)a->M, (long long)a->N, (long long)a->K, a->epilogue,
               a->residual ? "r" : "", a->c_dtype == DM_BF16 ? "h" : "f", tile);
      pname = shaped;
    }
    const double mn = (double)a->M * a->N;
    DmProfScope prof(pname, s, 2.0 * a->M * a->N * a->K,
                     esz * ((double)a->M * a->K + (double)a->N * a->K) + csz * mn * (a->accumulate ? 2.0 : 1.0) +
                         (a->residual ? 4.0 * mn : 0.0) + (a->aux ? ((a->aux_dtype == DM_BF16) ? 2.0 : 4.0) * mn : 0.0));
    const bool cs_t64 = !big && !w4 && !ring && a->layout == DM_TN && a->ab_dtype == DM_BF16 && tile == 64 && a->colsum_a != nullptr && split <= 128;
    p.colsum_slab = ((big || (w4 && a->layout == DM_TN) || cs_t64) && cs_region) ? cs_region : nullptr;
    if (w4) {
      dm_gemm_w4_launch(p, a->layout, w4, s);
    } else if (ring) {
      dm_gemm_ring_launch(p, ring, s);
    } else if (big) {
      dm_gemm256_launch(p, a->layout, s);
    } else if (a->ab_dtype == DM_BF16) {
      if (tile == 128) launch_mfma<bf16_t, 4>(p, a->layout, grid, s); else launch_mfma<bf16_t, 2>(p, a->layout, grid, s);
    } else {
      if (tile == 128) launch_mfma<float, 4>(p, a->layout, grid, s); else launch_mfma<float, 2>(p, a->layout, grid, s);
    }
    if (fwd_split && split > 1) {      // (inside the profiler scope: the class time of these products includes their reduction)
      const long long n8 = (long long)a->M * a->N / 8;
      const long long want = (n8 + 255) / 256;
      GemmParams q = p;
      q.split_k = 1;
      hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, s, q, p.workspace, split);
    }
  }
  DM_LAUNCH_CHECK("dm_gemm");
  if (fwd_split) return DM_OK;
  const bool cs_t64 = !big && !w4 && !ring && a->layout == DM_TN && a->ab_dtype == DM_BF16 && tile == 64 && a->colsum_a != nullptr && split <= 128;
  const bool cs_fused = big || (w4 && a->layout == DM_TN) || cs_t64;      // these kernels produce the partial column sums of A themselves
  const int cs_rows_per_slice = big ? 4 : 1;      // (256x256 pipeline: one row per wave column; 64x64 tiles and the 4-wave kernel: one per slice)
  if (split > 1) {
    const long long n4 = (long long)a->M * a->N / 4;
    const long long want = (n4 + 255) / 256;
    const int rgrid = (int)(want < 2048 ? want : 2048);
    const bool fold_cs = a->colsum_a && cs_fused;     // the pipelines' [split * 4 | 2][M] partial column sums ride along
    const int cs_blocks = fold_cs ? (a->M + 255) / 256 : 0;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rgrid + cs_blocks), dim3(256), 0, s, p.workspace,
                       reinterpret_cast<float *>(a->C), (long long)a->ldc, a->M, a->N, split, a->accumulate, rgrid,
                       fold_cs ? cs_region : nullptr, fold_cs ? a->colsum_a : nullptr, a->M, split * cs_rows_per_slice, a->colsum_accumulate);
    DM_LAUNCH_CHECK("dm_gemm(split-k reduce)");
    if (fold_cs) return DM_OK;
  }
  if (a->colsum_a) {
    if (cs_fused) {      // fold the pipeline's partial rows, in row order
      hipLaunchKernelGGL(colsum_rows_reduce_kernel, dim3((a->M + 63) / 64), dim3(64), 0, s, cs_region, a->colsum_a, a->M, split * cs_rows_per_slice,
                         a->colsum_accumulate);
      DM_LAUNCH_CHECK("dm_gemm(colsum reduce)");
    } else if (folded) {      // the distinct pieces of A, one pass each: colsum(hi) + colsum(lo)
      int accum = a->colsum_accumulate;
      for (int sgm = 0; sgm < 3; ++sgm) {
        bool first = true;
        for (int e = 0; e < sgm; ++e) first = first && a->a_fold[e] != a->a_fold[sgm];
        if (!first) continue;
        const int rc = dm_colsum(reinterpret_cast<const unsigned short *>(a->A) + a->a_fold[sgm], a->ab_dtype, a->lda, a->colsum_a, a->k_fold, a->M, accum, cs_region, stream);
        if (rc != DM_OK) return rc;
        accum = 1;
      }
    } else {
      const int rc = dm_colsum(a->A, a->ab_dtype, a->lda, a->colsum_a, a->K, a->M, a->colsum_accumulate, cs_region, stream);
      if (rc != DM_OK) return rc;
    }
  }
  return DM_OK;
}

// n independent products, results as n dm_gemm calls in order would give them (weight gradients: up to the order of the fp32 additions
// over K).  The products must not overlap in their outputs and none may read another's output.  Fast path: 1 .. 8 bf16 weight gradients
// (DM_TN, plain operands or plane pairs, fp32 C, no epilogue operands, whole 256 x 192 tiles, automatic slice count) in ONE launch of the
// 4-wave kernel (dm_gemm_w4_grouped: one K slice per tile for short contractions, stream-K + one fix-up launch for long ones; the column sums
// of A from the same launches).  Anything else: the calls one after the other.
extern "C" int64_t dm_gemm_grouped_workspace_bytes(const DmGemmArgs *args, int32_t n) {
  if (args == nullptr || n < 1 || n > 8) return 0;
  const char *menv = getenv("DM_GEMM_GROUPED");      // only the stream-K form (never chosen by the default rule) takes a group workspace
  if (!menv || atoi(menv) != 3) return 0;
  for (int i = 0; i < n; ++i)
    if (args[i].layout != DM_TN || args[i].ab_dtype != DM_BF16 || args[i].M % 256 != 0 || args[i].N % 192 != 0) return 0;
  return dm_gemm_w4_grouped_ws_bytes();
}

extern "C" int dm_gemm_grouped(const DmGemmArgs *args, int32_t n, void *workspace, int64_t workspace_bytes, void *stream) {
  DM_REQUIRE(args != nullptr && n >= 1, DM_ERR_BAD_SHAPE, "dm_gemm_grouped: null args / n < 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  bool fast = n <= 8;
  GemmParams ps[8];
  float *cs_rows[8] = {nullptr};       // form 1, per product: the row the launch writes when the sums must be ADDED to colsum_a afterwards
  DmGroupedExtra x{};
  x.ws = (workspace && dm_aligned16(workspace)) ? workspace : nullptr;
  x.ws_bytes = x.ws ? workspace_bytes : 0;
  for (int i = 0; fast && i < n; ++i) {
    const DmGemmArgs &a = args[i];
    fast = a.layout == DM_TN && a.ab_dtype == DM_BF16 && a.c_dtype == DM_F32 && a.epilogue == DM_EPI_NONE && a.split_k == 0 && a.k_fold >= 0 &&
           a.A && a.B && a.C && !a.bias && !a.residual && !a.aux && a.rows_per_group == 0 && a.M > 0 && a.N > 0 && a.K > 0 &&
           a.M % 8 == 0 && a.N % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 4 == 0 && a.lda >= a.M && a.ldb >= a.N && a.ldc >= a.N &&
           dm_aligned16(a.A) && dm_aligned16(a.B) && dm_aligned16(a.C);
    if (!fast) break;
    GemmParams p{};
    p.A = a.A; p.B = a.B; p.C = a.C;
    p.lda = a.lda; p.ldb = a.ldb; p.ldc = a.ldc; p.ldr = a.ldc; p.ldaux = a.ldc;
    p.M = a.M; p.N = a.N; p.K = a.K;
    p.epilogue = DM_EPI_NONE; p.accumulate = a.accumulate ? 1 : 0; p.c_dtype = DM_F32; p.aux_dtype = DM_F32;
    p.group_m = 8;
    if (a.k_fold > 0) {      // hi / lo plane pairs: dm_gemm's own preconditions, then the segments as given
      fast = a.K == 3 * a.k_fold && a.k_fold % 64 == 0;
      for (int sgm = 0; fast && sgm < 3; ++sgm) {
        fast = a.a_fold[sgm] >= 0 && a.a_fold[sgm] < (1LL << 30) && a.b_fold[sgm] >= 0 && a.b_fold[sgm] < (1LL << 30) && a.a_fold[sgm] % 8 == 0 && a.b_fold[sgm] % 8 == 0;
        p.a_fold[sgm] = a.a_fold[sgm];
        p.b_fold[sgm] = a.b_fold[sgm];
      }
      if (!fast) break;
      p.k_fold = a.k_fold;
    }
    x.cs_out[i] = a.colsum_a;
    x.cs_acc[i] = a.colsum_accumulate ? 1 : 0;
    {      // the product's own workspace, cut as dm_gemm cuts it: split-K slab, then the column-sum rows (the sliced form of the group)
      int64_t sb = a.workspace ? a.workspace_bytes : 0;
      x.cs_region[i] = nullptr;
      if (a.colsum_a && a.workspace) {
        const int64_t need = colsum_region_floats(a.M) * 4;
        if (a.workspace_bytes >= need) {
          sb = (a.workspace_bytes - need) & ~15LL;
          x.cs_region[i] = reinterpret_cast<float *>(reinterpret_cast<char *>(a.workspace) + sb);
        } else sb = 0;
      }
      x.slab[i] = sb > 0 ? reinterpret_cast<float *>(a.workspace) : nullptr;
      x.slab_bytes[i] = sb;
    }
    if (a.colsum_a) {
      if (!a.colsum_accumulate) {
        p.colsum_slab = a.colsum_a;                      // form 1, first write of the step: the launch stores the sums where they belong
      } else {
        const int64_t need = colsum_region_floats(a.M) * 4;
        fast = a.workspace != nullptr && a.workspace_bytes >= need;
        if (!fast) break;
        cs_rows[i] = reinterpret_cast<float *>(reinterpret_cast<char *>(a.workspace) + ((a.workspace_bytes - need) & ~15LL));
        p.colsum_slab = cs_rows[i];
      }
    }
    ps[i] = p;
  }
  const int form = fast ? dm_gemm_w4_grouped(ps, n, s, false, x) : 0;
  if (form != 0) {
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
      flops += 2.0 * args[i].M * args[i].N * args[i].K;
      bytes += 2.0 * ((double)args[i].M * args[i].K + (double)args[i].N * args[i].K) + 4.0 * (double)args[i].M * args[i].N * (args[i].accumulate ? 2.0 : 1.0);
    }
    static const bool by_shape = [] { const char *e = getenv("DM_PROF_SHAPES"); return e && e[0] == '1'; }();
    char shaped[64];
    const char *pname = "gemm_bf16_TN";
    if (by_shape) {
      snprintf(shaped, sizeof(shaped), "gemm_bf16_TN_grouped%d_K%d_%s", n, args[0].K, form == 2 ? "streamk" : form == 3 ? "sliced" : "1slice");
      pname = shaped;
    }
    {
      DmProfScope prof(pname, s, flops, bytes);      // (stream-K: the fix-up launch is inside the scope, like the reductions of sliced products are not)
      dm_gemm_w4_grouped(ps, n, s, true, x);
    }
    if (form == 1)
      for (int i = 0; i < n; ++i)
        if (cs_rows[i])
          hipLaunchKernelGGL(colsum_rows_reduce_kernel, dim3((args[i].M + 63) / 64), dim3(64), 0, s, cs_rows[i], args[i].colsum_a, args[i].M, 1, 1);
    if (form == 3)      // every product's slices summed in slice order, its column-sum rows folded in the same launch (as dm_gemm does)
      for (int i = 0; i < n; ++i) {
        const DmGemmArgs &a = args[i];
        const int split = ps[i].split_k;
        const long long want = ((long long)a.M * a.N / 4 + 255) / 256;
        const int rgrid = (int)(want < 2048 ? want : 2048);
        const bool fold_cs = a.colsum_a != nullptr;
        const int cs_blocks = fold_cs ? (a.M + 255) / 256 : 0;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rgrid + cs_blocks), dim3(256), 0, s, ps[i].workspace, reinterpret_cast<float *>(a.C), (long long)a.ldc,
                           a.M, a.N, split, a.accumulate, rgrid, fold_cs ? x.cs_region[i] : nullptr, fold_cs ? a.colsum_a : nullptr, a.M, split,
                           a.colsum_accumulate);
      }
    DM_LAUNCH_CHECK("dm_gemm_grouped");
    return DM_OK;
  }
  for (int i = 0; i < n; ++i) {
    const int rc = dm_gemm(&args[i], stream);
    if (rc != DM_OK) return rc;
  }
  return DM_OK;
}
