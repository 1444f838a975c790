// fp8 arithmetic mode (KD_PREC_FP8; BASELINE configs[4] "fp8 MFMA weights"): the AdaRMSNorm -> wide projections of the K = 256 / 512 levels
// (qkv + cosine-sim + RoPE, up projection + GEGLU, plain store) on gfx950's block-scaled fp8 matrix instruction,
// v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate, 5 PFLOP/s dense).  There is no reference counterpart
// (convert_for_inference.py:23 stops at fp16 / bf16); the arithmetic is defined HERE and restated in oracle/hdit.py (mx8_*):
//
//   weights      W[n, :] = 2^e_n * q_n,  q_n OCP e4m3 (RNE), e_n = ceil(log2(max|W[n, :]| / 448)): one power-of-two scale per output
//                channel -- exactly checkpoint.quantize_fp8, so an fp8 checkpoint's weights enter the instruction bit for bit;
//   activations  u = x * s (x the bf16 residual row, s the sample's fp32 AdaRMSNorm scale vector) is quantised per (row, 32-k block):
//                u[b] = 2^E_b * q_b, E_b = ceil(log2(max|u[b]| / 448)) from the fp32 quotient's bits (no saturation: |q| <= 448), q OCP e4m3
//                (RNE) -- OCP microscaling, which the instruction takes natively: one E8M0 scale byte per (row, 32-k block);
//   products     exact in the instruction, fp32 accumulation; the RMS row factor rsqrt(mean(x^2) + eps) (fp32 statistics of the UNQUANTISED
//                row) multiplies the accumulators in the epilogue, which is gemm_bf16.hip's (cosine-sim norm, RoPE, GELU in fp32).
//
// Kernel form: gemm_bf16.hip's A-stationary projection (gemm_astat_kernel) with the matrix instruction and the operand formats changed:
// a wave keeps its 32 rows as fp8 B fragments in registers (K / 8 registers instead of K / 4: both widths leave room for two waves per SIMD),
// the packed fp8 weight streams through the same 4-slot ring of 16 KiB blocks -- a block now covers 128 k instead of 64 -- by LDS-DMA, and a
// k-step of 64 is ONE instruction per 32-feature group (8 per block, 64 cycles each) fed by two ds_read_b128.
//
// Packed image (kd_pack_weight_mx8): blocks [n-tile][k / 128] of 16 KiB, inside a block the 16-byte piece (feature group j of 32, k-step ks
// of 64, 32-k block h of the step = half h of a lane's 32 bytes, lane half lh = which 16 k of that block, feature l31) at
// ((((j * 2 + ks) * 2 + h) * 2 + lh) * 32 + l31) * 16: every ds_read_b128 of a W fragment half covers one contiguous KiB.  Behind the blocks: one E8M0 byte per tile row, [n-tile][l31][j].
// GEGLU tiles interleave 32 value rows with their 32 gate rows like the bf16 image (bf16_common.h: w_row_of_tile).
#include "bf16_common.h"

namespace kd {
namespace b16 {
extern unsigned long long* g_clk;     // gemm_bf16.hip
}
namespace mx8 {
using namespace b16;

using i32x8 = __attribute__((ext_vector_type(8))) int;

struct MArgs {
  const u16* A; const char* Wp; const unsigned* Ws; u16* C;
  const float* scale; int scale_stride, rows_per_sample; float eps;
  int M, N, n_tiles, n_splits;
  int n_heads; const float* qk_scale; const float* pos; const float* freq;
  int warm;
};

// E8M0 exponent byte of the power-of-two block scale 2^(E - 127) >= amax / 448 (smallest such), clamped to [1, 253]
__device__ __host__ __forceinline__ unsigned mx_scale_byte(float amax) {
  const float r = amax / 448.0f;                 // correctly rounded: 448 * 2^n / 448 == 2^n
  unsigned b = __builtin_bit_cast(unsigned, r);
  b = (b + 0x7FFFFFu) >> 23;                     // exponent, + 1 unless the mantissa is zero
  return b < 1u ? 1u : (b > 253u ? 253u : b);
}
__device__ __forceinline__ float mx_inv_scale(unsigned byte) { return __uint_as_float((254u - byte) << 23); }

__device__ __forceinline__ void wait_vm_dyn8(int n) {
  switch (n) {
#define KD_C(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
    KD_C(0) KD_C(1) KD_C(2) KD_C(3) KD_C(4) KD_C(5) KD_C(6) KD_C(7) KD_C(8) KD_C(9) KD_C(10) KD_C(11) KD_C(12) KD_C(13) KD_C(14) KD_C(15)
    KD_C(16) KD_C(17) KD_C(18) KD_C(19) KD_C(20) KD_C(21) KD_C(22) KD_C(23) KD_C(24) KD_C(25) KD_C(26) KD_C(27) KD_C(28) KD_C(29) KD_C(30) KD_C(31)
#undef KD_C
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// NK64 = K / 64 (4 or 8).  128-row panels (4 waves), two workgroups per CU, n-splits of a panel on one XCD: gemm_astat_kernel's schedule.
template <int NK64, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_mx8_astat_kernel(const MArgs p) {
  constexpr int K = NK64 * 64, NKB = K / 128, NSTG = 4, PDIST = NSTG - 1, NWV = 4, PB = 4;
  constexpr bool GEGLU = EPI == KD_EPI_GEGLU;
  constexpr int NCOL = GEGLU ? 64 : 128;
  constexpr int NST = GEGLU ? 4 : 8;                 // 16-byte stores per lane per n-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const auto warm = code_warm_begin<16 * 1024>((int)blockIdx.x < p.warm && tid < 64);
  int panel, split;
  const int n_splits = p.n_splits, n_panels = gridDim.x / n_splits;
  if ((n_panels & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    panel = (j / n_splits) * 8 + xcd;
    split = j % n_splits;
  } else {
    panel = blockIdx.x % n_panels;
    split = blockIdx.x / n_panels;
  }
  const int nt_begin = (int)((long)p.n_tiles * split / n_splits), nt_end = (int)((long)p.n_tiles * (split + 1) / n_splits);
  const int n_tiles = nt_end - nt_begin, total = n_tiles * NKB;
  const int m0 = panel * 128;

  const char* wp = p.Wp + (size_t)nt_begin * NKB * WBLK + wid * (PB * 1024) + lane * 16;
  auto issue = [&](int s) {
    const char* src = wp + (size_t)s * WBLK;
    char* dst = smem + (s % NSTG) * WBLK + wid * (PB * 1024);
#pragma unroll
    for (int j = 0; j < PB; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024),
                                       (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
  };

  const int row = m0 + wid * 32 + l31;
  const bool ok = row < p.M;
  const int rowc = ok ? row : p.M - 1;
  // The instruction's operand layout, found by experiment (benchmarks/probe/mx8_probe.cpp, profiles/r06_mx8_probe.log): lane (row, lh) holds,
  // of a 64-wide k-step, bytes 0-15 = k 16 lh .. 16 lh + 15 of the step's FIRST 32-k block and bytes 16-31 = the same k of its SECOND block;
  // block b's E8M0 byte is read from the scale register of lane (row, lh = b); op_sel picks the byte of that register.
  i32x8 a[NK64];                                       // this lane's 2 x 16 e4m3 values of every 64-wide k-step
  int asc[NK64];                                       // ... and the E8M0 byte of block lh of the step, in all four byte lanes (any op_sel reads it)
  float rs;
  {
    // rows: HBM -> the ring slot this wave borrows (whole rows by LDS-DMA) -> this lane's 64-byte pieces; the sample's scale vector -> LDS
    constexpr int RPR = WBLK / (2 * K);                // rows per staging round: 16 at K = 512, 32 at K = 256
    constexpr int NR = 32 / RPR, CPR = K / 8;
    static_assert(RPR >= 16 && NR * RPR == 32, "staging geometry");
    u32x4 raw[NK64][4];
    char* stage = smem + wid * WBLK;
    char* scl = smem + NSTG * WBLK + wid * (K * 4);
    const int r_first = min(m0 + wid * 32, p.M - 1), r_last = min(m0 + wid * 32 + 31, p.M - 1);
    const bool uni = r_first / p.rows_per_sample == r_last / p.rows_per_sample;
    const int b = rowc / p.rows_per_sample;
    const float* sp = p.scale + (size_t)b * p.scale_stride;
    if (uni) {
      const char* ssrc = reinterpret_cast<const char*>(p.scale + (size_t)(r_first / p.rows_per_sample) * p.scale_stride) + lane * 16;
#pragma unroll
      for (int i = 0; i < K * 4 / 1024; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ssrc + i * 1024),
                                         (__attribute__((address_space(3))) void*)(scl + i * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i * 64 + lane) / CPR, qs = (i * 64 + lane) % CPR;
        const int grow = min(m0 + wid * 32 + r * RPR + rr, p.M - 1);
        const char* src = reinterpret_cast<const char*>(p.A + (size_t)grow * K) + ((qs ^ (rr & 15)) << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + i * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // wave-private slot: no barrier
      if (NR == 1 || (l31 / RPR) == r) {
        const int rr = l31 % RPR;
        const char* rowp = stage + rr * (2 * K);
#pragma unroll
        for (int ks = 0; ks < NK64; ++ks)
#pragma unroll
          for (int u = 0; u < 4; ++u)       // u = 2 * block + piece: 8 k of block (u >> 1), k = 16 lh + 8 (u & 1) .. inside it
            raw[ks][u] = *reinterpret_cast<const u32x4*>(rowp + (((8 * ks + 4 * (u >> 1) + 2 * lh + (u & 1)) ^ (rr & 15)) << 4));
      }
      if (NR > 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float ssq = 0.f;
    const float* spl = reinterpret_cast<const float*>(scl);
#pragma unroll
    for (int ks = 0; ks < NK64; ++ks) {
      float y[32];
      float amax[2] = {0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k0 = 64 * ks + 32 * (u >> 1) + 16 * lh + 8 * (u & 1);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>((uni ? spl : sp) + k0), s1 = *reinterpret_cast<const f32x4*>((uni ? spl : sp) + k0 + 4);
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[2 * e] = bf_lo(raw[ks][u][e]); x[2 * e + 1] = bf_hi(raw[ks][u][e]); }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ssq = fmaf(x[e], x[e], ssq);
          const float v = x[e] * (e < 4 ? s0[e] : s1[e - 4]);
          y[8 * u + e] = v;
          amax[u >> 1] = fmaxf(amax[u >> 1], fabsf(v));
        }
      }
      // a 32-k block lives in BOTH lanes of a row (16 values each): its maximum is the larger of the two halves'
      unsigned sb[2];
      float inv[2];
#pragma unroll
      for (int bk = 0; bk < 2; ++bk) {
        amax[bk] = fmaxf(amax[bk], __shfl_xor(amax[bk], 32, 64));
        sb[bk] = mx_scale_byte(amax[bk]);
        inv[bk] = mx_inv_scale(sb[bk]);
      }
      i32x8 f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        int pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(y[4 * w] * inv[w >> 2], y[4 * w + 1] * inv[w >> 2], pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(y[4 * w + 2] * inv[w >> 2], y[4 * w + 3] * inv[w >> 2], pk, true);
        f[w] = pk;
      }
      asm volatile("" : "+v"(f));                      // materialise the fragment here (see gemm_astat_kernel)
      a[ks] = f;
      asc[ks] = (int)((lh ? sb[1] : sb[0]) * 0x01010101u);      // block b's byte is read from the lane half b of the row
      __builtin_amdgcn_sched_barrier(0);
    }
    ssq += __shfl_xor(ssq, 32, 64);
    rs = rsqrtf(ssq / (float)K + p.eps);
  }
  float py = 0.f, px = 0.f;
  if (EPI == KD_EPI_QKV) {
    const int tok = rowc % p.rows_per_sample;
    py = p.pos[2 * tok];
    px = p.pos[2 * tok + 1];
  }
  code_warm_end(warm);
  // the E8M0 channel-scale bytes of this workgroup's n-tiles -> LDS (one dword per (tile, l31): the four feature groups' bytes)
  unsigned* wsl = reinterpret_cast<unsigned*>(smem + NSTG * WBLK + NWV * K * 4);
  for (int i = tid; i < n_tiles * 32; i += 256) wsl[i] = p.Ws[(size_t)nt_begin * 32 + i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");              // every wave has taken its rows out of the slot it borrowed; the scale bytes are in
#pragma unroll
  for (int s = 0; s < PDIST; ++s)
    if (s < total) issue(s);
  const bool full_panel = m0 + 128 <= p.M;
  u16* crow = p.C + (size_t)rowc * p.N;
  const int rd = lh * 512 + l31 * 16;                  // this lane's 16 bytes inside a (j, ks, h) KiB of a block

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  for (int nt = 0; nt < n_tiles; ++nt) {
    // the tile's four E8M0 channel-scale bytes of this lane's W rows (feature groups j = 0..3), each spread over a register's byte lanes
    const unsigned wsw = wsl[nt * 32 + l31];
    int wsc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wsc[j] = (int)(((wsw >> (8 * j)) & 0xFFu) * 0x01010101u);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int s = nt * NKB + kb;
      {
        int allow = PB * min(PDIST - 1, total - 1 - s);
        if (full_panel) {
          if (nt > 0 && kb + 1 <= PDIST) allow += NST;
          if (nt > 1 && kb + 1 + NKB <= PDIST) allow += NST;
        }
        wait_vm_dyn8(allow);
      }
      asm volatile("s_barrier" ::: "memory");
      if (s + PDIST < total) issue(s + PDIST);
      const char* st = smem + (s % NSTG) * WBLK + rd;
      i32x8 wf[2][4];
      auto read4 = [&](int ks, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4 h0 = *reinterpret_cast<const u32x4*>(st + ((j * 2 + ks) * 2 + 0) * 1024);
          const u32x4 h1 = *reinterpret_cast<const u32x4*>(st + ((j * 2 + ks) * 2 + 1) * 1024);
          wf[buf][j] = i32x8{(int)h0[0], (int)h0[1], (int)h0[2], (int)h0[3], (int)h1[0], (int)h1[1], (int)h1[2], (int)h1[3]};
        }
      };
      read4(0, 0);
      read4(1, 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[ks][j], a[2 * kb + ks], acc[j], 0, 0, 0, wsc[j], 0, asc[2 * kb + ks]);
      }
    }
    const int n0 = (nt_begin + nt) * NCOL;
    if (GEGLU) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        float v[16];
        const float rsh = 0.5f * rs;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 o = geglu_pair(f32x2{acc[2 * jj][r], acc[2 * jj][r + 1]} * rsh, f32x2{acc[2 * jj + 1][r], acc[2 * jj + 1][r + 1]} * rs);
          v[r] = o.x;
          v[r + 1] = o.y;
        }
        store_block_bf16(crow + n0 + 32 * jj, v, lh, ok);
      }
    } else if (EPI == KD_EPI_QKV) {
#pragma unroll
      for (int vv = 0; vv < 2; ++vv) {
        const int vec = (n0 >> 6) + vv;
        const int which = vec / p.n_heads, head = vec - which * p.n_heads;
        if (which < 2) {
          typedef float f32x8s __attribute__((ext_vector_type(8)));
          f32x8s fq;
          float qsc;
          asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                       : "=s"(fq), "=s"(qsc) : "s"(p.freq + head * 8), "s"(p.qk_scale + head) : "memory");
          float fr[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) fr[u] = pick_half(fq[u], fq[4 + u], 0u - (unsigned)lh);
          qk_prep_blocks(acc[2 * vv], acc[2 * vv + 1], rs, sqrtf(qsc), p.eps, py, px, fr);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[2 * vv][r] *= rs; acc[2 * vv + 1][r] *= rs; }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[2 * vv + jj][r];
          store_block_bf16(crow + n0 + 64 * vv + 32 * jj, v, lh, ok);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[j][r] * rs;
        store_block_bf16(crow + n0 + 32 * j, v, lh, ok);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  }
}

// ---- one-off weight packing: fp32 W [N, K] (GEGLU: [2 N, K]) -> e4m3 blocks + E8M0 channel scales -----------------------------------------
// One wave per tile row: amax over K, the power-of-two channel scale (checkpoint.quantize_fp8's rule), RNE conversion.
__global__ __launch_bounds__(64) void pack_weight_mx8_kernel(const float* __restrict__ W, char* __restrict__ out, int N, int K, int geglu, int n_tiles) {
  const int tr = blockIdx.x, nt = tr >> 7, r = tr & 127, lane = threadIdx.x;
  const int src = w_row_of_tile(nt, r, N, geglu != 0);
  const int nkb = K / 128;
  float amax = 0.f;
  if (src >= 0)
    for (int k = lane; k < K; k += 64) amax = fmaxf(amax, fabsf(W[(size_t)src * K + k]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const unsigned sb = mx_scale_byte(amax);
  const float inv = mx_inv_scale(sb);
  const int j = r >> 5, l31 = r & 31;
  for (int k4 = lane * 4; k4 < K; k4 += 256) {          // 4 consecutive k = one dword of a lane's fragment
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (src >= 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = W[(size_t)src * K + k4 + e] * inv;
    }
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
    const int kb = k4 >> 7, ks = (k4 >> 6) & 1, h = (k4 >> 5) & 1, lh = (k4 >> 4) & 1, byte = k4 & 15;      // h = 32-k block of the step, lh = its half
    const size_t off = ((size_t)nt * nkb + kb) * WBLK + ((((j * 2 + ks) * 2 + h) * 2 + lh) * 32 + l31) * 16 + byte;
    *reinterpret_cast<int*>(out + off) = pk;
  }
  if (lane == 0) out[(size_t)n_tiles * nkb * WBLK + (size_t)nt * 128 + l31 * 4 + j] = (char)sb;
}

}  // namespace mx8
}  // namespace kd

using namespace kd;
using namespace kd::mx8;

static int mx8_tiles(int N, int geglu) { return geglu ? (N + 63) / 64 : (N + 127) / 128; }

extern "C" long long kd_packed_weight_bytes_mx8(int N, int K, int geglu) {
  if (N <= 0 || K <= 0 || K % 128) return -1;
  const long long nt = mx8_tiles(N, geglu);
  return nt * (K / 128) * WBLK + nt * 128;
}

extern "C" int kd_pack_weight_mx8(const float* W, void* out, int N, int K, int geglu, void* stream) {
  if (!W || !out) return fail(KD_EINVAL, "kd_pack_weight_mx8: null operand");
  if (N <= 0 || K <= 0 || K % 128) return fail(KD_EINVAL, "kd_pack_weight_mx8: N=%d K=%d (K must be a multiple of 128)", N, K);
  const int nt = mx8_tiles(N, geglu);
  hipLaunchKernelGGL(pack_weight_mx8_kernel, dim3((unsigned)(nt * 128)), dim3(64), 0, (hipStream_t)stream, W, reinterpret_cast<char*>(out), N, K, geglu, nt);
  return check_launch("kd_pack_weight_mx8");
}

// AdaRMSNorm -> projection on the block-scaled fp8 matrix instruction.  `d` is the projection's descriptor as kd_gemm_bf16 takes it (bf16
// activations in and out, norm = 1, epi = KD_EPI_STORE / KD_EPI_QKV / KD_EPI_GEGLU, precision = KD_PREC_BF16) except that Wp points at the
// kd_pack_weight_mx8 image.  K in {256, 512}, N a multiple of the tile (128; GEGLU: 64), M >= 128.
extern "C" int kd_gemm_mx8_supported(int M, int N, int K, int epi, int norm) {
  if (!norm || (K != 256 && K != 512) || M < 128 || !option("mx8", 1)) return 0;
  if (epi == KD_EPI_GEGLU) return N > 0 && N % 64 == 0;
  if (epi == KD_EPI_QKV) return N == 3 * K;
  if (epi == KD_EPI_STORE) return N > 0 && N % 128 == 0;
  return 0;
}

extern "C" int kd_gemm_mx8(const KdGemm* dp, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_gemm_mx8: null descriptor");
  const KdGemm& d = *dp;
  if (!d.A || !d.Wp || !d.C || !d.scale) return fail(KD_EINVAL, "kd_gemm_mx8: null operand");
  if (d.a_mode != KD_A_PLAIN || d.precision != KD_PREC_BF16 || d.rows_per_sample <= 0 || (d.epi == KD_EPI_STORE && d.out_add != 0.f))
    return fail(KD_EINVAL, "kd_gemm_mx8: the descriptor must be a bf16 norm -> projection (plain A, rows_per_sample set)");
  if (!kd_gemm_mx8_supported(d.M, d.N, d.K, d.epi, d.norm))
    return fail(KD_EINVAL, "kd_gemm_mx8: shape M=%d N=%d K=%d epi=%d norm=%d is not taken (norm -> store / qkv / GEGLU, K in {256, 512})", d.M, d.N, d.K, d.epi, d.norm);
  if (d.epi == KD_EPI_QKV && (!d.qk_scale || !d.rope_pos || !d.rope_freq || d.n_heads * 64 != d.K))
    return fail(KD_EINVAL, "kd_gemm_mx8: the qkv projection needs qk_scale, rope_pos, rope_freq and n_heads * 64 == K");
  const bool geglu = d.epi == KD_EPI_GEGLU;
  const int n_tiles = mx8_tiles(d.N, geglu), nkb = d.K / 128;
  MArgs a{reinterpret_cast<const u16*>(d.A), reinterpret_cast<const char*>(d.Wp),
          reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(d.Wp) + (size_t)n_tiles * nkb * WBLK), reinterpret_cast<u16*>(d.C),
          d.scale, d.scale_stride, d.rows_per_sample, d.eps, d.M, d.N, n_tiles, 1,
          d.n_heads, d.qk_scale, d.rope_pos, d.rope_freq, option("code_warm", KD_CODE_WARM_DEFAULT)};
  // n-splits of a panel: gemm_astat's cost model (rounds x (row prologue + tiles per split)), two workgroups per CU
  const int panels = (d.M + 127) / 128, slots = 2 * cu_count();
  int best = 1;
  long best_cost = -1;
  for (int sp = 1; sp <= n_tiles; ++sp) {
    if (n_tiles % sp) continue;
    const long rounds = ((long)panels * sp + slots - 1) / slots;
    const long cost = rounds * (2 + n_tiles / sp);
    if (best_cost < 0 || cost < best_cost) { best = sp; best_cost = cost; }
  }
  const int forced = option("mx8_splits", 0);
  a.n_splits = forced > 0 && forced <= n_tiles ? forced : best;
  if ((n_tiles + a.n_splits - 1) / a.n_splits > 64) return fail(KD_EINVAL, "kd_gemm_mx8: more than 64 n-tiles per workgroup (N=%d)", d.N);
  hipStream_t s = (hipStream_t)stream;
  const double n_eff = geglu ? 2.0 * d.N : (double)d.N;
  const double flops = 2.0 * d.M * n_eff * d.K;
  const double bytes = 2.0 * ((double)d.M * d.K + (double)d.M * d.N) + n_eff * d.K;
  char nm[96] = "gemm_mx8_astat";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_mx8_astat<e%d> M=%d N=%d K=%d", d.epi, d.M, d.N, d.K);
  LaunchScope prof(nm, flops, bytes, s);
  // ring + one scale vector (K floats) per wave + the channel-scale bytes of a split's n-tiles (<= 64 tiles)
#define KD_MX(NKV, EP) { constexpr int LDS = 4 * WBLK + 4 * NKV * 64 * 4 + 64 * 128; static LdsAttr set;               \
    set.ensure(reinterpret_cast<const void*>(gemm_mx8_astat_kernel<NKV, EP>), LDS);                                      \
    hipLaunchKernelGGL((gemm_mx8_astat_kernel<NKV, EP>), dim3((unsigned)(panels * a.n_splits)), dim3(256), LDS, s, a); }
  if (d.K == 512) {
    if (d.epi == KD_EPI_GEGLU) KD_MX(8, KD_EPI_GEGLU) else if (d.epi == KD_EPI_QKV) KD_MX(8, KD_EPI_QKV) else KD_MX(8, KD_EPI_STORE)
  } else {
    if (d.epi == KD_EPI_GEGLU) KD_MX(4, KD_EPI_GEGLU) else if (d.epi == KD_EPI_QKV) KD_MX(4, KD_EPI_QKV) else KD_MX(4, KD_EPI_STORE)
  }
#undef KD_MX
  return check_launch("kd_gemm_mx8");
}

KD_TEXT_PAD(gemm_mx8)      // last function of this code object: kd_common.h, code warm-up
