// hoic_mlp.hip — the dense policy / value GEMMs of the PPO update on the matrix cores at float32 accuracy.
//
// The update (AgentPPO.update_policy / AgentPG.update_value, uhc/khrylib/rl/agents/agent_ppo.py:16-56, agent_pg.py:18-25)
// is 5 full-batch epochs of forward + backward through two 617-2048-1024-512 GELU MLPs on >= 50 000 samples: 12 TFLOP
// of float32 GEMM per iteration, 73 % of the loop at the f32 MFMA rate (157 TFLOP/s).  gfx950's f16 MFMA runs 16x
// faster.  Here every float32 operand x is carried as an error-free pair of halves
//     x * 2^e = hi + lo,   hi = f16(x 2^e),  lo = f16(x 2^e - hi)          (22 significand bits; e per tensor)
// and a product sum is three f16 MFMAs into ONE float32 accumulator:  hi.hi + hi.lo + lo.hi   (lo.lo < 2^-22 dropped).
// f16 x f16 products are exact in float32, so the result carries float32-class rounding (2^-22 per operand instead of
// 2^-24) at 16/3 of the f32 MFMA rate.  Power-of-two scales 2^e keep every tensor inside the f16 range and are undone
// exactly in the epilogue.
//
// Storage format "H8L8": a packed tensor [R x C] is R rows of 2C halves; columns come in groups of eight,
// [h0 .. h7 l0 .. l7] (32 bytes).  A lane's MFMA fragment (8 consecutive k) is then ONE 16-byte read for the hi halves
// and one for the lo halves, straight into the operand registers; an epilogue lane that owns 4 consecutive columns
// writes two 8-byte pieces.
//
// One GEMM kernel, "NT" form:  C[m][n] = alpha * sum_k A[m][k] B[n][k]  with A [M x K], B [N x K] packed, K contiguous.
//   forward      A = activations [M x K_in],  B = W [N_out x K_in]            epilogue: + bias, GELU, GELU', pack
//   backward dX  A = dZ [M x N_out],          B = W^T [K_in x N_out]          epilogue: * GELU'(z), pack (+ transposed)
//   backward dW  A = dZ^T [N_out x M],        B = H^T [K_in x M], split over M epilogue: float32 slabs
// Transposed copies of activations / gradients are written by the producing epilogue.
// Tile 256 x BN (BN = 256 | 128), 8 wavefronts, K step 32, two LDS stages filled by global_load_lds (16 B per lane,
// LDS image lane-linear, XOR swizzle applied to the SOURCE address and to the ds_read), v_mfma_f32_32x32x16_f16.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <string>
#include "../../include/hoic.h"
#include "hoic_zfilter_core.h"

extern "C" const char* hoic_last_error(void);
void hoic_set_error(const std::string& s);    // hoic_capi.hip
#define MCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { hoic_set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return HOIC_ERR_DEVICE; } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((__attribute__((address_space(1))) const void*)(p))

// ---------------------------------------------------------------------------------------------- split helpers
__device__ __forceinline__ void split_f16(float y, _Float16& h, _Float16& l) {
  h = (_Float16)y;
  l = (_Float16)(y - (float)h);
}
__device__ __forceinline__ unsigned pack_hl(float y) {      // hi in the low half-word, lo in the high one
  _Float16 h, l;
  split_f16(y, h, l);
  return (unsigned)__builtin_bit_cast(u16, h) | ((unsigned)__builtin_bit_cast(u16, l) << 16);
}
// GELU (exact erf form, torch.nn.GELU default) and its derivative from ONE exponential:
//   Phi(z) = 1/2 (1 + erf(z / sqrt 2)),  phi(z) = exp(-z^2 / 2) / sqrt(2 pi),  gelu = z Phi,  gelu' = Phi + z phi
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute) on t = 1 / (1 + p |x|), sharing exp(-x^2) with phi.
__device__ __forceinline__ void gelu_pair(float z, float& g, float& dg) {
  const float x = z * 0.70710678118654752f, ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  const float e = __expf(-x * x);                                     // = exp(-z^2 / 2)
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float erf_abs = fmaf(-poly, e, 1.f);
  const float Phi = 0.5f * (1.f + copysignf(erf_abs, x));
  g = z * Phi;
  dg = fmaf(z * 0.39894228040143268f, e, Phi);
}
// The same pair for TWO values at once on the packed float32 pipe (v_pk_fma_f32 / v_pk_mul_f32: two IEEE operations per
// instruction).  Constants folded: t = 1 / (1 + (p / sqrt 2) |z|), exp(-z^2 / 2) = exp2(-(z sqrt(log2(e) / 2))^2).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_pair2(f2 z, f2& g, f2& dg) {
  const f2 one = {1.f, 1.f}, half = {0.5f, 0.5f};
  const f2 az = {fabsf(z.x), fabsf(z.y)};
  const f2 den = __builtin_elementwise_fma(az, (f2){0.23164190f, 0.23164190f}, one);
  const f2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  const f2 u = z * (f2){0.84932180028801904f, 0.84932180028801904f};
  const f2 mu2 = -(u * u);
  const f2 e = {__builtin_amdgcn_exp2f(mu2.x), __builtin_amdgcn_exp2f(mu2.y)};
  f2 p = __builtin_elementwise_fma(t, (f2){1.061405429f, 1.061405429f}, (f2){-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(t, p, (f2){1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(t, p, (f2){-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(t, p, (f2){0.254829592f, 0.254829592f});
  p = p * t;
  const f2 ea = __builtin_elementwise_fma(-p, e, one);
  const f2 s = {copysignf(ea.x, z.x), copysignf(ea.y, z.y)};
  const f2 Phi = __builtin_elementwise_fma(s, half, half);
  g = z * Phi;
  dg = __builtin_elementwise_fma(z * (f2){0.39894228040143268f, 0.39894228040143268f}, e, Phi);
}
// hi | lo << 16 of y * so (so a power of two: the product is exact), as two v_fma_mix
__device__ __forceinline__ unsigned pack_hl_scaled(float y, float so) {
  h2 X;
  X.x = (_Float16)(y * so);
  X.y = (_Float16)(y * so - (float)X.x);
  return __builtin_bit_cast(unsigned, X);
}
__device__ __forceinline__ float wave_max_f(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---------------------------------------------------------------------------------------------- the GEMM kernel
enum { EPI_F32 = 0, EPI_FWD = 1, EPI_BWD = 2 };
struct GemmArgs {
  const u16* A; const u16* B;     // packed H8L8: A [M x 2K] halves, B [N x 2K] halves
  int M, N, K;                    // K = contraction length in elements (multiple of 32)
  int kt_per_split;               // K stages (of 32) handled by one block (split-K over blockIdx.y)
  const int* exps;                // device: scale exponents of the tensor slots
  float* amax;                    // device: running max |value| per tensor slot (float bits, atomicMax on uint)
  int ea, eb, eo;                 // slots of A, B and of the packed output
  float extra_scale;              // multiplies alpha (e.g. 1 / loss scale is folded into the exponents instead; 1.0)
  // EPI_F32
  float* C; long long c_split_stride;     // [M x N] float32 (+ slab stride per split)
  // EPI_FWD / EPI_BWD
  const float* bias;              // [N] (FWD)
  const float* Gin;               // [M x N] float32: GELU'(z) of the layer whose pre-activation gradient is formed (BWD)
  float* Gout;                    // [M x N] float32 GELU'(z) (FWD)
  float* Hf32;                    // optional [M x N] float32 copy of the output
  u16* P;                         // optional packed output [M x 2N]
  u16* PT;                        // optional packed transposed output [N x 2M]
  float* colpart;                 // optional (BWD, D[m][n] epilogue): [M / 128][N] column sums of the output per 128-row chunk
};

template <int BN> struct Cfg {
  static constexpr int WM = (BN == 256) ? 2 : 4;       // wavefronts along m / n
  static constexpr int WN = 8 / WM;
  static constexpr int TM = 256 / WM / 32;               // 32 x 32 MFMA tiles per wavefront along m / n
  static constexpr int TN = BN / WN / 32;
  static constexpr int STAGE = (256 + BN) * 128;         // bytes per LDS stage
};

// stage one K step (32 elements = 128 bytes per row) of `rows` rows into LDS: chunk q (16 B) of the image belongs to
// row q >> 3, and holds source chunk (q & 7) ^ ((row >> 1) & 7) of that row's 128-byte segment
template <int ROWS> __device__ __forceinline__ void stage_rows(const char* g, size_t rowbytes, char* lds, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < ROWS / 64; i++) {
    const int q = (i * 8 + wave) * 64 + lane, row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
    __builtin_amdgcn_global_load_lds(GLB_PTR(g + (size_t)row * rowbytes + c * 16), LDS_PTR(lds + (i * 8 + wave) * 1024), 16, 0, 0);
  }
}
// hi and lo halves of k = 16 s + 8 hf ... + 7 of `row`: group g = 2 s + hf of the row's four groups, chunks 2g and 2g + 1
__device__ __forceinline__ void read_frag(const char* tile, int row, int s, int hf, h8& hi, h8& lo) {
  const int sw = (row >> 1) & 7, c = 2 * (2 * s + hf);
  hi = *(const h8*)(tile + row * 128 + ((c ^ sw) << 4));
  lo = *(const h8*)(tile + row * 128 + (((c + 1) ^ sw) << 4));
}

// ---------------------------------------------------------------------------------------------- epilogue
// acc[i][j][reg] of a wavefront whose tile starts at (mb, nb): m = mb + 32 i + (lane & 31),
// n = nb + 32 j + 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3): every lane owns runs of 4 consecutive n of one m
template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16 (&acc)[TM][TN], int mb, int nb, int split, int lane) {
  const int l31 = lane & 31, hf = lane >> 5;
  const int ea = a.exps ? a.exps[a.ea] : 0, eb = a.exps ? a.exps[a.eb] : 0;
  const float alpha = ldexpf(a.extra_scale, -(ea + eb));
  if (EPI == EPI_F32) {
    float* Cp = a.C + (size_t)split * a.c_split_stride;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int m = mb + 32 * i + l31, n = nb + 32 * j + 8 * rg + 4 * hf;
          f32x4 v = {alpha * acc[i][j][4 * rg], alpha * acc[i][j][4 * rg + 1], alpha * acc[i][j][4 * rg + 2], alpha * acc[i][j][4 * rg + 3]};
          *(f32x4*)(Cp + (size_t)m * a.N + n) = v;
        }
    return;
  }
  const int eo = a.exps ? a.exps[a.eo] : 0;
  const float so = ldexpf(1.f, eo);
  float vmax = 0.f;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int m = mb + 32 * i + l31, n = nb + 32 * j + 8 * rg + 4 * hf;
        float v[4];
        if (EPI == EPI_FWD) {
          const f32x4 b4 = *(const f32x4*)(a.bias + n);
          f32x4 dg4;
#pragma unroll
          for (int r = 0; r < 4; r++) { float gq, dq; gelu_pair(fmaf(alpha, acc[i][j][4 * rg + r], b4[r]), gq, dq); v[r] = gq; dg4[r] = dq; }
          if (a.Gout) *(f32x4*)(a.Gout + (size_t)m * a.N + n) = dg4;
        } else {
          const f32x4 g4 = *(const f32x4*)(a.Gin + (size_t)m * a.N + n);
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = alpha * acc[i][j][4 * rg + r] * g4[r];
        }
        if (a.Hf32) { const f32x4 o = {v[0], v[1], v[2], v[3]}; *(f32x4*)(a.Hf32 + (size_t)m * a.N + n) = o; }
        unsigned w[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { vmax = fmaxf(vmax, fabsf(v[r])); w[r] = pack_hl(v[r] * so); }
        if (a.P) {      // 4 of the 8 columns of a group: 8 bytes of hi halves, 8 bytes of lo halves 16 bytes further
          u16* g8 = a.P + ((size_t)m * a.N + (n & ~7)) * 2 + (n & 4);
          const u32x2 oh = {(w[0] & 0xffffu) | (w[1] << 16), (w[2] & 0xffffu) | (w[3] << 16)};
          const u32x2 ol = {(w[0] >> 16) | (w[1] & 0xffff0000u), (w[2] >> 16) | (w[3] & 0xffff0000u)};
          *(u32x2*)g8 = oh; *(u32x2*)(g8 + 8) = ol;
        }
        if (a.PT) {     // transposed: 4 consecutive m of a column n are spread over a quad of lanes; lane q writes one dword:
                        // q = 0: (h0,h1)  1: (h2,h3)  2: (l0,l1)  3: (l2,l3) of the quad's half of the 8-group
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int wa = __builtin_amdgcn_update_dpp(0, (int)w[r], 0x88, 0xf, 0xf, false);   // quad_perm [0,2,0,2]
            const int wb = __builtin_amdgcn_update_dpp(0, (int)w[r], 0xdd, 0xf, 0xf, false);   // quad_perm [1,3,1,3]
            const unsigned o = __builtin_amdgcn_perm((unsigned)wb, (unsigned)wa, (lane & 2) ? 0x07060302u : 0x05040100u);
            *(unsigned*)(a.PT + ((size_t)(n + r) * a.M + (m & ~7)) * 2 + ((lane & 2) ? 8 : 0) + (m & 4) + ((lane & 1) ? 2 : 0)) = o;
          }
        }
      }
  if (a.amax) {
    vmax = wave_max_f(vmax);
    if (lane == 0) atomicMax((unsigned*)(a.amax + a.eo), __float_as_uint(vmax));
  }
}

// The same epilogue for accumulators in the OTHER orientation, D[m][n] (activation-side fragment = MFMA A operand): a
// lane owns column n = nb + 32 j + (lane & 31) and, per register, row m = mb + 32 i + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// Every global access is then one dword per lane with 32 lanes on 128 consecutive bytes (full lines): GELU' and the
// float32 copy directly, the packed row-major output after an exchange inside each group of 8 lanes (lane q of a group
// writes dword q of the 32-byte H8L8 group: q < 4 the hi halves of columns 2q, 2q + 1, q >= 4 the lo halves).  No
// transposed output in this form (the weight-gradient kernel below reads row-major operands).
// Round 6: the arithmetic runs two rows at a time on the packed float32 pipe (v_pk_fma_f32 / v_pk_mul_f32: consecutive
// accumulator registers are consecutive rows of one column), the hi | lo word of an element is two v_fma_mix (the output
// scale folded in), and every access is base (SGPR pair) + one 32-bit byte offset shared by the output arrays -- 21 issue
// slots per element where the scalar form took 41 (profiles/r06_experiments.json "epilogue_diet").
template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_mn(const GemmArgs& a, f32x16 (&acc)[TM][TN], int mb, int nb, int lane) {
  const int l31 = lane & 31, hf = lane >> 5;
  const int ea = a.exps ? a.exps[a.ea] : 0, eb = a.exps ? a.exps[a.eb] : 0;
  const float alpha = ldexpf(a.extra_scale, -(ea + eb));
  const int eo = a.exps ? a.exps[a.eo] : 0;
  const float so = ldexpf(1.f, eo);
  const int src0 = ((lane & ~7) + 2 * (lane & 3)) << 2, src1 = src0 + 4;        // ds_bpermute byte addresses of the two source lanes
  const unsigned sel = (lane & 4) ? 0x07060302u : 0x05040100u;                  // lanes 4..7 of a group assemble the lo halves
  // addresses: the arrays' base pointers stay in scalar registers, the BYTE offset of an element is one 32-bit VGPR (outputs are
  // below 2^30 elements: checked in hoic_mlp_gemm), shared by the packed output, GELU' and the float32 copy (all 4 bytes per element)
  const unsigned N4 = (unsigned)a.N * 4u;
  const unsigned lane_ob = (unsigned)mb * N4 + (unsigned)(nb + l31) * 4u + 4u * (unsigned)hf * N4;
  float vmax = 0.f;
  const char* __restrict__ Gin = (const char*)a.Gin;
  char* __restrict__ Gout = (char*)a.Gout;
  char* __restrict__ Hf = (char*)a.Hf32;
  char* __restrict__ P32 = (char*)a.P;
  const f2 alpha2 = {alpha, alpha};
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const float bj = (EPI == EPI_FWD) ? a.bias[nb + 32 * j + l31] : 0.f;
    const f2 bj2 = {bj, bj};
    float csum = 0.f;      // this lane's column over the wavefront's 128 rows (its half of them): the bias gradient's partial sum
#pragma unroll
    for (int i = 0; i < TM; i++) {
      // half a 32 x 32 tile at a time, in batches: 8 loads in flight, then the arithmetic, then 16 lane exchanges in flight,
      // then the stores (element by element every access would wait for the one before it)
#pragma unroll
      for (int hb = 0; hb < 2; hb++) {
        __builtin_amdgcn_sched_barrier(0);
        unsigned o[8];
        f2 v[4], gq[4];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int r = 8 * hb + q;
          o[q] = lane_ob + (unsigned)(32 * i + (r & 3) + 8 * (r >> 2)) * N4 + 128u * j;
          if (EPI == EPI_BWD) gq[q >> 1][q & 1] = *(const float*)(Gin + o[q]);
        }
        if (EPI == EPI_BWD) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const f2 ac = {acc[i][j][8 * hb + 2 * p], acc[i][j][8 * hb + 2 * p + 1]};
          if (EPI == EPI_FWD) gelu_pair2(__builtin_elementwise_fma(alpha2, ac, bj2), v[p], gq[p]);
          else v[p] = (alpha2 * ac) * gq[p];
          vmax = fmaxf(vmax, fmaxf(fabsf(v[p].x), fabsf(v[p].y)));
          if (EPI == EPI_BWD) { csum += v[p].x; csum += v[p].y; }
        }
        if (P32) {
          unsigned w0[8], w1[8];
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const unsigned w = pack_hl_scaled(v[q >> 1][q & 1], so);
            w0[q] = (unsigned)__builtin_amdgcn_ds_bpermute(src0, (int)w); w1[q] = (unsigned)__builtin_amdgcn_ds_bpermute(src1, (int)w);
          }
#pragma unroll
          for (int q = 0; q < 8; q++) *(unsigned*)(P32 + o[q]) = __builtin_amdgcn_perm(w1[q], w0[q], sel);
        }
        if (EPI == EPI_FWD && Gout) {
#pragma unroll
          for (int q = 0; q < 8; q++) *(float*)(Gout + o[q]) = gq[q >> 1][q & 1];
        }
        if (Hf) {
#pragma unroll
          for (int q = 0; q < 8; q++) *(float*)(Hf + o[q]) = v[q >> 1][q & 1];
        }
      }
    }
    if (EPI == EPI_BWD && a.colpart) {      // TM * 32 = 128 rows per wavefront: chunk mb / 128, fixed order => deterministic
      static_assert(TM == 4, "column partials are per 128-row chunk");
      csum += __shfl_xor(csum, 32);
      if (hf == 0) a.colpart[(size_t)(mb >> 7) * a.N + nb + 32 * j + l31] = csum;
    }
  }
  if (a.amax) {
    vmax = wave_max_f(vmax);
    if (lane == 0) atomicMax((unsigned*)(a.amax + a.eo), __float_as_uint(vmax));
  }
}

// (the 8-wavefront two-stage kernel of the first f16x3 version was superseded by the 4-wavefront K16 kernels below and removed)

// ---------------------------------------------------------------------------------------------- second tiling
// 4 wavefronts, tile 256 x 128 (wavefront tile 128 x 64 as above), K stages of 16, THREE LDS buffers of 24 KB: two
// workgroups fit on a CU (2 x 72 KB LDS, one wavefront of each per SIMD).  While one workgroup sits in its epilogue
// (HBM-write bound: packed output, its transpose and GELU' are 12 bytes per element) or at a barrier, the other one
// keeps the matrix cores busy; with one 8-wavefront workgroup per CU nothing overlaps the epilogue.
// Pipeline per stage t: [wait for the LDS-DMA of stage t+1 with a COUNTED vmcnt (stage t+2's may stay in flight), raw
// s_barrier] -> issue the DMA of stage t+3 into the buffer stage t was read from -> LDS reads of stage t+1's fragments
// -> 24 MFMAs on stage t's fragments (already in registers).
#define K16_STG ((256 + 128) * 64)
// ---- main loop shared by the 4-wavefront kernels (forward / data gradient: hoic_gemm_f16x3_k16_kernel, weight gradient:
// hoic_gemm_f16x3_tn16_kernel).  The kernel defines K16_RD(F, q, stage) = LDS reads of fragment q (0, 1: B; 2..5: A) and
// K16_DMA(piece, source stage, buffer offset) = one LDS-DMA piece (6 per wavefront and stage), declares acc[4][2], Frags,
// smem, nkt (EVEN: K and the split sizes are multiples of 32) and MN, then expands K16_MAINLOOP.
// MFMA k of a step (0..23): product k / 8 (hi.lo, lo.hi, hi.hi), tile (k % 8) / 2, (k % 8) % 2 -- eight different
// accumulators in a row, so no MFMA waits for the one before it
#define K16_MF(Fx, KK)                                                                                                  \
  {                                                                                                                     \
    constexpr int p_ = (KK) / 8, i_ = ((KK) % 8) / 2, j_ = (KK) % 2;                                                    \
    const h8 av_ = (p_ == 1) ? Fx.al[i_] : Fx.ah[i_], bv_ = (p_ == 0) ? Fx.bl[j_] : Fx.bh[j_];                            \
    if (MN) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av_, bv_, acc[i_][j_], 0, 0, 0);                          \
    else acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv_, av_, acc[i_][j_], 0, 0, 0);                             \
  }
#define SB __builtin_amdgcn_sched_barrier(0);
#define K16_MMA_ALL(Fx)                                                                                                 \
  K16_MF(Fx, 0) K16_MF(Fx, 1) K16_MF(Fx, 2) K16_MF(Fx, 3) K16_MF(Fx, 4) K16_MF(Fx, 5) K16_MF(Fx, 6) K16_MF(Fx, 7)         \
  K16_MF(Fx, 8) K16_MF(Fx, 9) K16_MF(Fx, 10) K16_MF(Fx, 11) K16_MF(Fx, 12) K16_MF(Fx, 13) K16_MF(Fx, 14) K16_MF(Fx, 15)   \
  K16_MF(Fx, 16) K16_MF(Fx, 17) K16_MF(Fx, 18) K16_MF(Fx, 19) K16_MF(Fx, 20) K16_MF(Fx, 21) K16_MF(Fx, 22) K16_MF(Fx, 23)
// One step, written out instruction by instruction (a scheduling fence after each item, so the order below IS the issue
// order): the LDS reads of stage t + 1 and the DMA pieces of stage t + 3 (clamped to the last stage: a step past the end
// re-fetches it into a free buffer, so every step issues the same 6 pieces and one counted wait fits all) go out in the
// shadow of this wavefront's own MFMAs of stage t.
#define K16_STEP(Fc, Fn)                                                                                                \
    {                                                                                                                   \
      __builtin_amdgcn_s_waitcnt(0x0076);    /* vmcnt(6): stage t+1 landed (t+2's pieces may stay in flight) */         \
      __builtin_amdgcn_s_barrier();          /* ... everywhere; buffer t % 3 is read out */                             \
      SB                                                                                                                \
      const int src_ = min(t + 3, last);                                                                                \
      const int on_ = (ob == 2 * K16_STG) ? 0 : ob + K16_STG;      /* buffer of stage t + 1 */                             \
      const char* nx_ = smem + on_;                                                                                     \
      K16_MF(Fc, 0) SB K16_RD(Fn, 0, nx_) SB K16_MF(Fc, 1) SB K16_RD(Fn, 1, nx_) SB K16_MF(Fc, 2) SB K16_RD(Fn, 2, nx_) SB  \
      K16_MF(Fc, 3) SB K16_RD(Fn, 3, nx_) SB K16_MF(Fc, 4) SB K16_RD(Fn, 4, nx_) SB K16_MF(Fc, 5) SB K16_RD(Fn, 5, nx_) SB  \
      K16_MF(Fc, 6) SB K16_DMA(0, src_, ob) SB K16_MF(Fc, 7) K16_MF(Fc, 8) K16_MF(Fc, 9) SB                              \
      K16_DMA(1, src_, ob) SB K16_MF(Fc, 10) K16_MF(Fc, 11) K16_MF(Fc, 12) SB                                            \
      K16_DMA(2, src_, ob) SB K16_MF(Fc, 13) K16_MF(Fc, 14) K16_MF(Fc, 15) SB                                            \
      K16_DMA(3, src_, ob) SB K16_MF(Fc, 16) K16_MF(Fc, 17) K16_MF(Fc, 18) SB                                            \
      K16_DMA(4, src_, ob) SB K16_MF(Fc, 19) K16_MF(Fc, 20) K16_MF(Fc, 21) SB                                            \
      K16_DMA(5, src_, ob) SB K16_MF(Fc, 22) K16_MF(Fc, 23) SB                                                          \
      __builtin_amdgcn_s_waitcnt(0xc07f);    /* lgkmcnt(0): the next fragments arrived long ago */                      \
      t++; ob = on_;                                                                                                    \
    }
// nkt - 1 steps that prefetch, then the last stage's MFMAs -- ONE tail, so the accumulators keep their registers (two
// alternative tails met in a join that went through scratch)
#define K16_MAINLOOP                                                                                                    \
  const int last = nkt - 1;                                                                                             \
  if (nkt > 0) {                                                                                                        \
    _Pragma("unroll") for (int pc = 0; pc < 6; pc++) K16_DMA(pc, 0, 0)                                                  \
    _Pragma("unroll") for (int pc = 0; pc < 6; pc++) K16_DMA(pc, min(1, last), K16_STG)                                 \
    __builtin_amdgcn_s_waitcnt(0x0076);      /* vmcnt(6): stage 0 landed */                                             \
    __builtin_amdgcn_s_barrier();                                                                                       \
    _Pragma("unroll") for (int pc = 0; pc < 6; pc++) K16_DMA(pc, min(2, last), 2 * K16_STG)                             \
    Frags F0, F1;                                                                                                       \
    K16_RD(F0, 0, smem) K16_RD(F0, 1, smem) K16_RD(F0, 2, smem) K16_RD(F0, 3, smem) K16_RD(F0, 4, smem) K16_RD(F0, 5, smem) \
    __builtin_amdgcn_s_waitcnt(0xc07f);      /* lgkmcnt(0) */                                                           \
    int t = 0, ob = 0;         /* ob: byte offset of the LDS buffer that holds stage t (and receives stage t + 3) */    \
    while (t < last - 1) {     /* two steps per trip: the two fragment sets are addressed statically */                 \
      K16_STEP(F0, F1)                                                                                                  \
      K16_STEP(F1, F0)                                                                                                  \
    }                                                                                                                   \
    K16_STEP(F0, F1)                                                                                                    \
    K16_MMA_ALL(F1)                                                                                                     \
    __builtin_amdgcn_s_waitcnt(0x0070);      /* the clamped re-fetches of the last steps */                             \
  }
template <int ROWS> __device__ __forceinline__ void stage_rows16(const char* g, size_t rowbytes, char* lds, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < ROWS / 64; i++) {      // 64-byte rows (16 elements): chunk q belongs to row q >> 2, source chunk (q & 3) ^ ((row >> 2) & 3)
    const int q = (i * 4 + wave) * 64 + lane, row = q >> 2, c = (q & 3) ^ ((row >> 2) & 3);
    __builtin_amdgcn_global_load_lds(GLB_PTR(g + (size_t)row * rowbytes + c * 16), LDS_PTR(lds + (i * 4 + wave) * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ void read_frag16(const char* tile, int row, int hf, h8& hi, h8& lo) {
  const int sw = (row >> 2) & 3;
  hi = *(const h8*)(tile + row * 64 + (((2 * hf) ^ sw) << 4));
  lo = *(const h8*)(tile + row * 64 + (((2 * hf + 1) ^ sw) << 4));
}
template <int EPI, bool MN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void hoic_gemm_f16x3_k16_kernel(GemmArgs a) {
  constexpr int TM = 4, TN = 2;
  __shared__ __attribute__((aligned(1024))) char smem[3 * K16_STG];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int ntn = a.N / 128, ntiles = (a.M / 256) * ntn;
  int g = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = g & 7, idx = g >> 3;
    g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (g / ntn) * 256, n0 = (g % ntn) * 128;
  const int split = blockIdx.y;
  const int kt0 = split * a.kt_per_split * 2;                       // kt_per_split counts stages of 32
  const int nkt = min(a.kt_per_split * 2, a.K / 16 - kt0);
  const size_t rowbytes = (size_t)a.K * 4;
  const char* Ag = (const char*)a.A + (size_t)m0 * rowbytes + (size_t)kt0 * 64;
  const char* Bg = (const char*)a.B + (size_t)n0 * rowbytes + (size_t)kt0 * 64;
  const int mw = (wave >> 1) * 128, nw = (wave & 1) * 64;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  struct Frags { h8 ah[TM], al[TM], bh[TN], bl[TN]; };
  // LDS byte offsets of this lane's first A / B fragment inside a stage (row * 64 + swizzled chunk); the lo halves sit in the
  // neighbouring chunk (offset ^ 16), fragment q of an operand 32 rows = 2048 bytes further (same swizzle: 32 rows = 8 groups of 4)
  const int rowA = mw + l31, rowB = nw + l31;
  const int fA = rowA * 64 + (((2 * hf) ^ ((rowA >> 2) & 3)) << 4), fB = 256 * 64 + rowB * 64 + (((2 * hf) ^ ((rowB >> 2) & 3)) << 4);
  const int fA1 = fA ^ 16, fB1 = fB ^ 16;
  // source of DMA piece pc (0..3: A rows, 4, 5: B rows): piece i covers rows (4 i + wave) * 16 + lane / 4 -- everything but
  // the lane's own part is wave-uniform, so the address is a scalar base + ONE 32-bit per-lane offset shared by all pieces
  const unsigned voff = (unsigned)(lane >> 2) * (unsigned)rowbytes + (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  const size_t piece_rows = (size_t)16 * rowbytes;     // bytes between the row groups of consecutive wavefronts
#define K16_RD(Fx, Q, STG)                                                                                              \
  {                                                                                                                     \
    if ((Q) < TN) {                                                                                                     \
      Fx.bh[(Q) < TN ? (Q) : 0] = *(const h8*)((STG) + fB + (Q) * 2048); Fx.bl[(Q) < TN ? (Q) : 0] = *(const h8*)((STG) + fB1 + (Q) * 2048);         \
    } else {                                                                                                            \
      Fx.ah[(Q) >= TN ? (Q) - TN : 0] = *(const h8*)((STG) + fA + ((Q) - TN) * 2048);                                     \
      Fx.al[(Q) >= TN ? (Q) - TN : 0] = *(const h8*)((STG) + fA1 + ((Q) - TN) * 2048);                                    \
    }                                                                                                                   \
  }
#define K16_DMA(PC, SRC, BUF)                                                                                           \
  __builtin_amdgcn_global_load_lds(GLB_PTR(((PC) < 4 ? Ag : Bg) + ((size_t)((((PC) < 4 ? (PC) : (PC) - 4)) * 4 + wave)) * piece_rows + (size_t)(SRC) * 64 + voff), \
                                   LDS_PTR(smem + (BUF) + ((PC) < 4 ? 0 : 256 * 64) + ((((PC) < 4 ? (PC) : (PC) - 4)) * 4 + wave) * 1024), 16, 0, 0);
  K16_MAINLOOP
#undef K16_DMA
#undef K16_RD
  if (MN) gemm_epilogue_mn<EPI, TM, TN>(a, acc, m0 + mw, n0 + nw, lane);
  else gemm_epilogue<EPI, TM, TN>(a, acc, m0 + mw, n0 + nw, split, lane);
}

// ---------------------------------------------------------------------------------------------- weight gradients
// C[i][j] = alpha sum_m A[m][i] B[m][j]  with BOTH operands row-major over the contraction index m (A = dZ [rows x 2 NA],
// B = H [rows x 2 NB], packed along their columns): the activations and gradients are used as the forward / data-gradient
// epilogues wrote them, no transposed copies.  The MFMA wants 8 consecutive m per lane for one column; gfx950's
// ds_read_b64_tr_b16 delivers exactly that from a row-major LDS image: within 16 lanes, source lane 4a + b supplies the
// address of [row a][4 halves of column block b] and result lane c receives column c of rows 0..3 (measured with
// tools/probe/tr_probe.hip).  Tile 256 (i) x BN (j), 8 wavefronts, K stages of 32 rows, two LDS stages by LDS-DMA: a
// stage row is 1 KB (BN = 128: 512 B) and 16-byte chunk c of row r sits at chunk c ^ ((r & 1) | ((r & 2) << 2)), which
// puts the 32 lanes of a transposing read on 32 distinct 8-byte slots of the 256-byte bank row.  Output: float32 slabs
// [split][NA x NB], one dword per lane with 32 lanes on one 128-byte line.
typedef __fp16 hw4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
// (its 8-wavefront row-major predecessor hoic_gemm_f16x3_tn_kernel: removed, superseded by the kernel below)
// The same product on the 4-wavefront main loop (K16_MAINLOOP above): tile 256 (i) x 128 (j), stages of 16 samples (A part
// 16 rows x 1 KB, B part 16 rows x 512 B = the 24 KB of the other kernel's stage), three LDS buffers, two workgroups per CU,
// every DMA piece one contiguous 1 KB row (A) or two 512-byte rows (B).  Transposing reads: with the stage's swizzle
// (chunk ^ ((row & 1) | ((row & 2) << 2))) fragment q of an operand sits 128 (q ^ s3) bytes from fragment 0, s3 = bit 1 of the
// lane's row -- two base addresses per operand half (even / odd q) and immediates cover all fragments.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void hoic_gemm_f16x3_tn16_kernel(GemmArgs a) {
  constexpr int TM = 4, TN = 2, RA = 1024, RB = 512, BOFF = 16 * RA;
  constexpr bool MN = true;                       // D[i][j]: lane = column j, registers = rows i
  __shared__ __attribute__((aligned(1024))) char smem[3 * K16_STG];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int ntn = a.N / 128;
  const int lin = blockIdx.x + gridDim.x * blockIdx.y, nsplit = gridDim.y;      // split-major: see hoic_gemm_f16x3_tn_kernel
  const int split = lin % nsplit, g = lin / nsplit;
  const int i0 = (g / ntn) * 256, j0 = (g % ntn) * 128;
  const int kt0 = split * a.kt_per_split * 2;                       // kt_per_split counts 32 samples = 2 stages
  const int nkt = min(a.kt_per_split * 2, a.K / 16 - kt0);
  const size_t lda = (size_t)a.M * 4, ldb = (size_t)a.N * 4;       // bytes per sample row of the packed operands
  const char* Ag = (const char*)a.A + (size_t)kt0 * 16 * lda + (size_t)i0 * 4;
  const char* Bg = (const char*)a.B + (size_t)kt0 * 16 * ldb + (size_t)j0 * 4;
  const int wi = (wave >> 1) * 128, wj = (wave & 1) * 64;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  struct Frags { h8 ah[TM], al[TM], bh[TN], bl[TN]; };
  // transposing-read addresses (read_frag_tr's mapping with 16-row stages): lane group gq = lane / 16 covers columns
  // 16 (gq & 1) .. + 15 and rows 8 (gq / 2) .. + 7 of a 32-column fragment; lane il = lane % 16 addresses row il / 4 (+ 4 for the
  // second read), columns 4 (il & 3) .. + 3
  const int gq = lane >> 4, il = lane & 15;
  const int row0 = 8 * (gq >> 1) + (il >> 2), s0 = row0 & 1, s3 = (row0 >> 1) & 1;
  const int chl = 4 * (gq & 1) + 2 * ((il & 3) >> 1), ho = (il & 1) ? 8 : 0;
  const int aB0 = row0 * RA + 16 * (wi / 4 + (chl ^ s0)) + ho, aB1 = row0 * RA + 16 * (wi / 4 + ((chl + 1) ^ s0)) + ho;
  const int bB0 = BOFF + row0 * RB + 16 * (wj / 4 + (chl ^ s0)) + ho, bB1 = BOFF + row0 * RB + 16 * (wj / 4 + ((chl + 1) ^ s0)) + ho;
  const int aE0 = aB0 + 128 * s3, aO0 = aB0 + 128 * (1 - s3), aE1 = aB1 + 128 * s3, aO1 = aB1 + 128 * (1 - s3);
  const int bE0 = bB0 + 128 * s3, bO0 = bB0 + 128 * (1 - s3), bE1 = bB1 + 128 * s3, bO1 = bB1 + 128 * (1 - s3);
  typedef unsigned long long u64;
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define TR_RD(OFF) __builtin_bit_cast(u64, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hw4*)LDS_PTR(OFF)))
#define K16_RD(Fx, Q, STG)                                                                                              \
  {                                                                                                                     \
    if ((Q) < TN) {                                                                                                     \
      const char* h_ = (STG) + (((Q) & 1) ? bO0 : bE0) + 128 * ((Q) & ~1);                                                \
      const char* l_ = (STG) + (((Q) & 1) ? bO1 : bE1) + 128 * ((Q) & ~1);                                                \
      const u64x2 hv_ = {TR_RD(h_), TR_RD(h_ + 4 * RB)}, lv_ = {TR_RD(l_), TR_RD(l_ + 4 * RB)};                            \
      Fx.bh[(Q) < TN ? (Q) : 0] = __builtin_bit_cast(h8, hv_); Fx.bl[(Q) < TN ? (Q) : 0] = __builtin_bit_cast(h8, lv_);     \
    } else {                                                                                                            \
      constexpr int q_ = (Q) >= TN ? (Q) - TN : 0;                                                                      \
      const char* h_ = (STG) + ((q_ & 1) ? aO0 : aE0) + 128 * (q_ & ~1);                                                  \
      const char* l_ = (STG) + ((q_ & 1) ? aO1 : aE1) + 128 * (q_ & ~1);                                                  \
      const u64x2 hv_ = {TR_RD(h_), TR_RD(h_ + 4 * RA)}, lv_ = {TR_RD(l_), TR_RD(l_ + 4 * RA)};                            \
      Fx.ah[q_] = __builtin_bit_cast(h8, hv_); Fx.al[q_] = __builtin_bit_cast(h8, lv_);                                   \
    }                                                                                                                   \
  }
  // DMA pieces: 0..3 = A row 4 pc + wave (1 KB, lane = chunk), 4, 5 = B rows 8 (pc - 4) + 2 wave + lane / 32 (512 B each);
  // the source chunk is the LDS chunk ^ swizzle(row), and the row's low bits come from the wavefront (and lane / 32) only
  const unsigned voffA = (unsigned)((lane ^ ((wave & 1) | ((wave & 2) << 2))) << 4);
  const unsigned voffB = (unsigned)(lane >> 5) * (unsigned)ldb + (unsigned)((((lane & 31) ^ ((lane >> 5) | ((wave & 1) << 3)))) << 4);
#define K16_DMA(PC, SRC, BUF)                                                                                           \
  {                                                                                                                     \
    if ((PC) < 4) __builtin_amdgcn_global_load_lds(GLB_PTR(Ag + ((size_t)(SRC) * 16 + 4 * (PC) + wave) * lda + voffA),    \
                                                   LDS_PTR(smem + (BUF) + (4 * (PC) + wave) * RA), 16, 0, 0);             \
    else __builtin_amdgcn_global_load_lds(GLB_PTR(Bg + ((size_t)(SRC) * 16 + 8 * ((PC) - 4) + 2 * wave) * ldb + voffB),   \
                                          LDS_PTR(smem + (BUF) + BOFF + (8 * ((PC) - 4) + 2 * wave) * RB), 16, 0, 0);     \
  }
  K16_MAINLOOP
#undef K16_DMA
#undef K16_RD
#undef TR_RD
  const int ea = a.exps ? a.exps[a.ea] : 0, eb = a.exps ? a.exps[a.eb] : 0;
  const float alpha = ldexpf(a.extra_scale, -(ea + eb));
  float* __restrict__ Cp = a.C + (size_t)split * a.c_split_stride;
  const unsigned N = (unsigned)a.N, lane_off = (unsigned)(i0 + wi + 4 * hf) * N + (unsigned)(j0 + wj + l31);
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        Cp[lane_off + (unsigned)(32 * i + (r & 3) + 8 * (r >> 2)) * N + 32u * j] = alpha * acc[i][j][r];
}

// ---------------------------------------------------------------------------------------------- pack kernels
// four consecutive elements (hi | lo << 16 each) of an 8-group: dst = group base + (column & 4) halves
__device__ __forceinline__ void store_quad(u16* dst, const unsigned (&w)[4]) {
  const u32x2 oh = {(w[0] & 0xffffu) | (w[1] << 16), (w[2] & 0xffffu) | (w[3] << 16)};
  const u32x2 ol = {(w[0] >> 16) | (w[1] & 0xffff0000u), (w[2] >> 16) | (w[3] & 0xffff0000u)};
  *(u32x2*)dst = oh; *(u32x2*)(dst + 8) = ol;
}
// float32 [R x C] (row stride ld) -> packed [Rp x 2 Cp] (zero padded), scaled by 2^exps[slot]; one thread per 4 columns
__global__ void hoic_pack_rows_kernel(const float* __restrict__ x, int R, int Cc, long long ld, u16* __restrict__ P, int Rp, int Cp,
                                      const int* __restrict__ exps, int slot) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = Cp >> 2;
  if (gid >= (long long)Rp * groups) return;
  const int r = (int)(gid / groups), c = (int)(gid % groups) * 4;
  const float s = ldexpf(1.f, exps ? exps[slot] : 0);
  unsigned w[4];
#pragma unroll
  for (int k = 0; k < 4; k++) w[k] = pack_hl((r < R && c + k < Cc) ? x[(long long)r * ld + c + k] * s : 0.f);
  store_quad(P + ((long long)r * Cp + (c & ~7)) * 2 + (c & 4), w);
}
// float32 [R x C] -> packed TRANSPOSE [Cp x 2 Rp]; optional elementwise factor y [R x C] (dZ = dH * GELU'); 64 x 64
// tiles through LDS.  grid (Cp / 64, Rp / 64), 256 threads.
__global__ __launch_bounds__(256) void hoic_pack_transpose_kernel(const float* __restrict__ x, const float* __restrict__ y, int R, int Cc, long long ld,
                                                                 u16* __restrict__ PT, int Rp, int Cp, const int* __restrict__ exps, int slot) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tid = threadIdx.x;
  const float s = ldexpf(1.f, exps ? exps[slot] : 0);
  for (int k = tid; k < 64 * 64; k += 256) {
    const int rr = k >> 6, cc = k & 63, r = r0 + rr, c = c0 + cc;
    float v = 0.f;
    if (r < R && c < Cc) { v = x[(long long)r * ld + c]; if (y) v *= y[(long long)r * ld + c]; }
    tile[rr][cc] = v * s;
  }
  __syncthreads();
  for (int k = tid; k < 64 * 16; k += 256) {       // output row c (64 of them), 16 groups of 4 r each
    const int cc = k >> 4, gq = (k & 15) * 4;
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) w[q] = pack_hl(tile[gq + q][cc]);
    store_quad(PT + ((long long)(c0 + cc) * Rp + ((r0 + gq) & ~7)) * 2 + ((r0 + gq) & 4), w);
  }
}
// dZ = dH * G (float32 [R x C], both row stride C) -> packed rows [Rp x 2C]; C multiple of 4
__global__ void hoic_pack_rows_mul_kernel(const float* __restrict__ x, const float* __restrict__ y, int R, int Cc, u16* __restrict__ P, int Rp,
                                          const int* __restrict__ exps, int slot) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = Cc >> 2;
  if (gid >= (long long)Rp * groups) return;
  const int r = (int)(gid / groups), c = (int)(gid % groups) * 4;
  const float s = ldexpf(1.f, exps ? exps[slot] : 0);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (r < R) { const f32x4 a = *(const f32x4*)(x + (long long)r * Cc + c), b = *(const f32x4*)(y + (long long)r * Cc + c); v = a * b; }
  unsigned w[4];
#pragma unroll
  for (int k = 0; k < 4; k++) w[k] = pack_hl(v[k] * s);
  store_quad(P + ((long long)r * Cc + (c & ~7)) * 2 + (c & 4), w);
}
// max |x * y| (y optional) over a float32 array -> amax[slot] (atomicMax on the float bits; values are non-negative)
__global__ void hoic_amax_kernel(const float* __restrict__ x, const float* __restrict__ y, long long n, float* __restrict__ amax, int slot) {
  float m = 0.f, nf = 0.f;       // nf: 0 while every element is finite (v * 0 is NaN for Inf and NaN; fmaxf alone would drop NaN)
  const long long n4 = ((((size_t)x | (size_t)(y ? y : x)) & 15) == 0) ? (n >> 2) : 0;      // 16-byte aligned: four elements per load
  const long long stride = (long long)gridDim.x * blockDim.x;
#pragma unroll 4
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 v = ((const f32x4*)x)[i];
    if (y) v = v * ((const f32x4*)y)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    nf = fmaf(v[0], 0.f, fmaf(v[1], 0.f, fmaf(v[2], 0.f, fmaf(v[3], 0.f, nf))));
  }
  for (long long i = 4 * n4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v = x[i]; if (y) v *= y[i];
    m = fmaxf(m, fabsf(v)); nf = fmaf(v, 0.f, nf);
  }
  if (nf != 0.f) m = __builtin_inff();       // an Inf / NaN element: the slot's maximum becomes Inf and hoic_update_exps counts it
  m = wave_max_f(m);
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned*)(amax + slot), __float_as_uint(m));
}
// exponent of slot i from its running maximum: 2^e * amax lands in [2^(target-1), 2^target); amax == 0 keeps e; then the
// maximum is cleared for the next pass.  One thread per slot.  mask bit i set = slot i is updated.
__global__ void hoic_update_exps_kernel(int* __restrict__ exps, float* __restrict__ amax, int nslots, unsigned long long mask, int target,
                                        int exact, int* __restrict__ overflow) {
  const int i = threadIdx.x;
  if (i >= nslots || !((mask >> i) & 1ull)) return;
  const float m = amax[i];
  if (m > 0.f && isfinite(m)) {
    int ex; frexpf(m, &ex);             // m = f * 2^ex, f in [0.5, 1)
    if (!exact && ldexpf(m, exps[i]) > 60000.f && overflow) atomicAdd(overflow, 1);     // the pass just measured overflowed f16
    exps[i] = target - ex;
  } else if (!isfinite(m) && overflow) atomicAdd(overflow, 1);
  amax[i] = 0.f;
}
// sum of S float32 slabs [rows x cols] -> out (row stride ldo, only the first out_cols columns), scaled
__global__ void hoic_slab_reduce_kernel(const float* __restrict__ slabs, int S, long long stride, int rows, int cols, float* __restrict__ out,
                                        int out_cols, long long ldo, float scale) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)rows * out_cols) return;
  const int r = (int)(gid / out_cols), c = (int)(gid % out_cols);
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < S; k++) s += slabs[(long long)k * stride + (long long)r * cols + c];      // fixed order; the loads are independent
  out[(long long)r * ldo + c] = s * scale;
}
// row sums of a packed [rows x 2 Cp] tensor (hi + lo), unscaled by 2^-exps[slot]: the bias gradient from dZ^T.
// One 256-thread block per row, fixed summation order.
__global__ __launch_bounds__(256) void hoic_rowsum_packed_kernel(const u16* __restrict__ P, int Cp, float* __restrict__ out, const int* __restrict__ exps, int slot) {
  __shared__ float part[256];
  const int r = blockIdx.x, tid = threadIdx.x;
  const u32x4* row = (const u32x4*)(P + (long long)r * Cp * 2);
  float s = 0.f;
  for (int gq = tid; gq < (Cp >> 3); gq += 256) {      // one 8-group per trip: 16 bytes of hi halves + 16 bytes of lo halves
    const h8 h = __builtin_bit_cast(h8, row[2 * gq]), l = __builtin_bit_cast(h8, row[2 * gq + 1]);
#pragma unroll
    for (int k = 0; k < 8; k++) s += (float)h[k] + (float)l[k];
  }
  part[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) part[tid] += part[tid + o]; __syncthreads(); }
  if (tid == 0) out[r] = ldexpf(part[0], -(exps ? exps[slot] : 0));
}

// ---------------------------------------------------------------------------------------------- rollout forward
// The policy's forward pass DURING the rollout runs next to the simulator's substep kernel, whose workgroups hold every
// CU's LDS (12 x 12.5 KB) and, at three 168-register wavefronts on a SIMD, 504 of its 512 registers: a GEMM whose workgroups need
// LDS queues behind them (the hipBLASLt float32 GEMMs of the 2048-row batches took 6 ms of a 36 ms rollout that way).  This kernel
// needs NO LDS and 152 registers: its wavefronts start on every SIMD that holds at most two substep wavefronts (the room beside
// two is 176 registers) -- a third of the wave slots are empty over a rollout because a launch ends with its slowest env, which is
// where this kernel runs; on a SIMD with three substep wavefronts nothing else becomes resident.  Operands go from L2 straight
// into MFMA registers, which asks for a layout in which a wavefront's operand load is one contiguous kilobyte --
// "tiled" format T of a matrix [R x K]: tile (a, s) = rows 32 a .. + 31, k = 16 s .. + 15, 2 KB at ((a K/16 + s) * 2048):
//   hi plane [64 lanes x 16 B], then lo plane; lane = 32 hf + l31 holds row 32 a + l31, k = 16 s + 8 hf .. + 7.
// One wavefront per workgroup computes 32 rows (m) x 64 columns (n): D = W X^T per tile, i.e. lane = row m, registers =
// columns n = (r & 3) + 8 (r >> 2) + 4 hf -- after one v_permlane32_swap per register pair a lane holds 8 consecutive n,
// which IS the next layer's operand fragment, so the epilogue (bias + GELU) writes format T again, 1 KB per store.
__global__ __launch_bounds__(64) void hoic_pack_tiled_kernel(const float* __restrict__ x, int R, int Cc, long long ld, char* __restrict__ T, int Rp,
                                                             int Kp, const int* __restrict__ exps, int slot) {
  const long long gid = (long long)blockIdx.x * 64 + threadIdx.x;        // one lane slot of one tile
  const int ksteps = Kp >> 4;
  const long long tile = gid >> 6;
  if (tile >= (long long)(Rp >> 5) * ksteps) return;
  const int lane = (int)(gid & 63), l31 = lane & 31, hf = lane >> 5;
  const int a = (int)(tile / ksteps), st = (int)(tile % ksteps);
  const int r = 32 * a + l31, c0 = 16 * st + 8 * hf;
  const float sc = ldexpf(1.f, exps ? exps[slot] : 0);
  unsigned w[8];
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = pack_hl((r < R && c0 + k < Cc) ? x[(long long)r * ld + c0 + k] * sc : 0.f);
  u32x4 hi, lo;
#pragma unroll
  for (int k = 0; k < 4; k++) { hi[k] = (w[2 * k] & 0xffffu) | (w[2 * k + 1] << 16); lo[k] = (w[2 * k] >> 16) | (w[2 * k + 1] & 0xffff0000u); }
  char* t = T + tile * 2048 + lane * 16;
  *(u32x4*)t = hi; *(u32x4*)(t + 1024) = lo;
}

// ---- the sampler's observation filter WITH the forward's operand (round 5): the filter's second launch -- merge of the chunk
// moments, new filter state, normalisation of the range's observations (float32 rows for the batch) -- also writes the rows in format
// T for hoic_fwd_tiled_kernel and refreshes the engine's delayed exponents: four launches of a range's chain (filter moments,
// filter apply, hoic_update_exps, hoic_pack_tiled) become two.  The arithmetic is that of hoic_zfilter (hoic_zfilter_core.h: the
// same source), so states and filter are bit-identical.  A workgroup is one wavefront: 16 columns x 32 rows (one tile of format T).
// (A first version also recomputed the chunk moments per workgroup to be ONE launch: sixteen-fold re-reads in 64-byte pieces,
//  373 us against 58 us for the four launches it replaced.)
__global__ __launch_bounds__(64) void hoic_zfilter_moments2_kernel(const float* __restrict__ x, int n, int dim, double* __restrict__ partial) {
  const int col = blockIdx.x * 64 + threadIdx.x, chunk = blockIdx.y;
  if (col >= dim) return;
  const int r0 = chunk * ZF_ROWS, r1 = min(r0 + ZF_ROWS, n);
  double mean, m2;
  zf_chunk_moments(x, dim, col, r0, r1, mean, m2);
  partial[((size_t)chunk * dim + col) * 2] = mean;
  partial[((size_t)chunk * dim + col) * 2 + 1] = m2;
}
__global__ __launch_bounds__(64) void hoic_zfilter_tiled_kernel(const float* __restrict__ x, int n, int dim, const double* __restrict__ partial,
                                                                 const double* __restrict__ state_in, double* __restrict__ state_out, int update,
                                                                 float clip, float* __restrict__ y, char* __restrict__ T, int Kp,
                                                                 int* __restrict__ exps, int slot_x, float* __restrict__ amax, int nslots,
                                                                 unsigned long long mask, int target, int* __restrict__ overflow,
                                                                 u16* __restrict__ P, int KpP) {
  // ONE wavefront per workgroup (a 32-row tile x 16 columns), like every other kernel of a range's chain: beside three 168-register
  // substep wavefronts a SIMD has no room for another wavefront, and a workgroup of four must find four free slots on one CU at the
  // same time -- the 256-thread form of this kernel took 280 us of mostly waiting.
  __shared__ double s_mu[16], s_rd[16];
  const int tid = threadIdx.x, strip = blockIdx.x, rowblk = blockIdx.y;       // rowblk: tile row (32 rows)
  const int nchunk = n / ZF_ROWS;
  // hidden-activation exponents of the forward engine from the last pass's maxima (hoic_update_exps_kernel, delayed slots)
  if (strip == 0 && rowblk == 0 && tid < nslots && ((mask >> tid) & 1ull)) {
    const float m = amax[tid];
    if (m > 0.f && isfinite(m)) {
      int ex; frexpf(m, &ex);
      if (ldexpf(m, exps[tid]) > 60000.f && overflow) atomicAdd(overflow, 1);
      exps[tid] = target - ex;
    } else if (!isfinite(m) && overflow) atomicAdd(overflow, 1);
    amax[tid] = 0.f;
  }
  if (tid < 16) {
    const int col = strip * 16 + tid;
    double mu = 0.0, rd = 0.0;
    if (col < dim) {
      double cnt = state_in[0], mean = state_in[1 + col], S = state_in[1 + dim + col];
      if (update) {
        for (int k = 0; k < nchunk; k++) zf_merge(cnt, mean, S, (double)ZF_ROWS, partial[((size_t)k * dim + col) * 2], partial[((size_t)k * dim + col) * 2 + 1]);
        if (rowblk == 0) {
          state_out[1 + col] = mean; state_out[1 + dim + col] = S;
          if (col == 0) state_out[0] = cnt;
        }
      }
      mu = mean; rd = zf_rden(cnt, mean, S);
    }
    s_mu[tid] = mu; s_rd[tid] = rd;
  }
  __syncthreads();
  // normalise + pack: tile row a = rowblk, k step = strip; lane (l31, hf) = row 32 a + l31, columns 16 strip + 8 hf .. + 7 (the lane
  // mapping of hoic_pack_tiled_kernel)
  const int lane = tid, l31 = lane & 31, hf = lane >> 5;
  const int a = rowblk, r = 32 * a + l31, c0 = 16 * strip + 8 * hf;
  const float sc = ldexpf(1.f, exps[slot_x]);
  const double lim = (double)clip;
  unsigned w[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    float v = 0.f;
    if (c0 + k < dim) {
      v = zf_apply(x[(size_t)r * dim + c0 + k], s_mu[8 * hf + k], s_rd[8 * hf + k], lim);
      y[(size_t)r * dim + c0 + k] = v;
    }
    w[k] = pack_hl(v * sc);
  }
  u32x4 hi, lo;
#pragma unroll
  for (int k = 0; k < 4; k++) { hi[k] = (w[2 * k] & 0xffffu) | (w[2 * k + 1] << 16); lo[k] = (w[2 * k] >> 16) | (w[2 * k + 1] & 0xffff0000u); }
  char* t = T + ((size_t)a * (Kp >> 4) + strip) * 2048 + lane * 16;
  *(u32x4*)t = hi; *(u32x4*)(t + 1024) = lo;
  // the same eight columns as one H8L8 group of the UPDATE's row-major operand (hoic_mlp_pack, rows [n x 2 KpP]): the rollout
  // leaves the batch packed for the update's first-layer GEMMs, which then need neither a maximum pass nor a pack pass over
  // the 53 k x 617 states (same exponent: the filter's clip bounds the input)
  if (P) { u16* g8 = P + ((size_t)r * KpP + c0) * 2; *(u32x4*)g8 = hi; *(u32x4*)(g8 + 8) = lo; }
}

struct FwdArgs {
  const char* X; const char* W;      // format T: X [M x K], W [N x K]
  int M, N, K;
  const int* exps; float* amax; int ex, ew, eo;
  const float* bias;                 // [N]
  char* outT;                        // format T [M x N] at 2^exps[eo] (hidden layers), or
  float* outF;                       // float32 row-major [M x N] (last layer)
};
template <bool LAST>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(168))) void hoic_fwd_tiled_kernel(FwdArgs a) {
  const int lane = threadIdx.x, l31 = lane & 31, hf = lane >> 5;
  const int ksteps = a.K >> 4, nt2 = a.N >> 6;
  const int mt = blockIdx.x / nt2, n2 = blockIdx.x % nt2;          // 32-row tile, pair of 32-column tiles
  const char* xp = a.X + (size_t)mt * ksteps * 2048 + lane * 16;
  const char* w0 = a.W + (size_t)(2 * n2) * ksteps * 2048 + lane * 16;
  const char* w1 = w0 + (size_t)ksteps * 2048;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
  struct Fr { h8 xh, xl, ah, al, bh, bl; };
  auto ld = [&](int st, Fr& f) {
    const size_t o = (size_t)st * 2048;
    f.xh = *(const h8*)(xp + o); f.xl = *(const h8*)(xp + o + 1024);
    f.ah = *(const h8*)(w0 + o); f.al = *(const h8*)(w0 + o + 1024);
    f.bh = *(const h8*)(w1 + o); f.bl = *(const h8*)(w1 + o + 1024);
  };
  auto mm = [&](const Fr& f) {       // D[n][m]: first operand = weights
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah, f.xl, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh, f.xl, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al, f.xh, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl, f.xh, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah, f.xh, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh, f.xh, acc1, 0, 0, 0);
  };
  // three register sets: the loads of step t + 2 are issued before the MFMAs of step t, so an operand has two steps' MFMAs
  // (~400 cycles) to arrive from L2 (with two sets a step waited for most of its loads' latency: the kernel ran at a fifth of
  // its MFMA bound); a step past the end re-reads the last stage
  Fr f0, f1, f2;
  const int last = ksteps - 1;
  ld(0, f0); ld(min(1, last), f1);
  int st = 0;
  for (; st + 3 <= ksteps; st += 3) {
    ld(min(st + 2, last), f2); mm(f0);
    ld(min(st + 3, last), f0); mm(f1);
    ld(min(st + 4, last), f1); mm(f2);
  }
  if (st < ksteps) mm(f0);
  if (st + 1 < ksteps) mm(f1);
  // ---- epilogue
  const int ex = a.exps ? a.exps[a.ex] : 0, ew = a.exps ? a.exps[a.ew] : 0;
  const float alpha = ldexpf(1.f, -(ex + ew));
  float vmax = 0.f;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const f32x16& acc = j ? acc1 : acc0;
    const int nb = 64 * n2 + 32 * j;
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 b4 = *(const f32x4*)(a.bias + nb + 8 * q + 4 * hf);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float g_, d_;
        gelu_pair(fmaf(alpha, acc[4 * q + r], b4[r]), g_, d_);
        v[4 * q + r] = g_;
        vmax = fmaxf(vmax, fabsf(g_));
      }
    }
    if (LAST) {
      float* o = a.outF + (size_t)(32 * mt + l31) * a.N + nb + 4 * hf;
#pragma unroll
      for (int q = 0; q < 4; q++) { const f32x4 t4 = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; *(f32x4*)(o + 8 * q) = t4; }
    } else {
      const float so = ldexpf(1.f, a.exps ? a.exps[a.eo] : 0);
      unsigned w[16];
#pragma unroll
      for (int r = 0; r < 16; r++) w[r] = pack_hl(v[r] * so);
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {           // the two 16-column halves of this 32-column tile = two k steps of the next layer
        unsigned e[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u32x2 p = __builtin_amdgcn_permlane32_swap(w[8 * s2 + k], w[8 * s2 + 4 + k], false, false);
          e[k] = p[0]; e[4 + k] = p[1];          // lane hf 0: n 0..7 of the half, lane hf 1: n 8..15
        }
        u32x4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; k++) { hi[k] = (e[2 * k] & 0xffffu) | (e[2 * k + 1] << 16); lo[k] = (e[2 * k] >> 16) | (e[2 * k + 1] & 0xffff0000u); }
        char* t = a.outT + ((size_t)mt * (a.N >> 4) + (size_t)(nb >> 4) + s2) * 2048 + lane * 16;
        *(u32x4*)t = hi; *(u32x4*)(t + 1024) = lo;
      }
    }
  }
  if (!LAST && a.amax) {
    vmax = wave_max_f(vmax);
    if (lane == 0) atomicMax((unsigned*)(a.amax + a.eo), __float_as_uint(vmax));
  }
}

// ---------------------------------------------------------------------------------------------- C-ABI
// 2: the 4-wavefront 256 x 128 K16 kernel, two workgroups per CU, accumulators in D[n][m] orientation: forward / data-gradient
//    epilogues also write transposed copies and the weight gradients are EPI_F32 products of those (kept as the second,
//    independently laid out form the tests compare the default against);
// 3 (default): the accumulators in D[m][n] orientation for the forward / data-gradient epilogues (every store a full line;
//    no transposed outputs: the weight gradients come from the row-major kernel hoic_mlp_gemm_tn)
static int g_gemm_pipeline = 3;
extern "C" int32_t hoic_mlp_set_pipeline(int32_t mode) { g_gemm_pipeline = mode < 3 ? 2 : 3; return HOIC_OK; }

template <int BN, int EPI> static int32_t launch_gemm(const GemmArgs& a, int splits, hipStream_t st) {
  if (g_gemm_pipeline == 3 && EPI != EPI_F32 && !a.PT)
    hipLaunchKernelGGL((hoic_gemm_f16x3_k16_kernel<EPI, true>), dim3((a.M / 256) * (a.N / 128), splits), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((hoic_gemm_f16x3_k16_kernel<EPI, false>), dim3((a.M / 256) * (a.N / 128), splits), dim3(256), 0, st, a);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_mlp_gemm(int32_t epi, int32_t M, int32_t N, int32_t K, const void* d_A, const void* d_B, const int32_t* d_exps,
                                 float* d_amax, int32_t slot_a, int32_t slot_b, int32_t slot_out, float extra_scale, int32_t splits,
                                 float* d_C, const float* d_bias, const float* d_gin, float* d_gout, float* d_hf32, void* d_P, void* d_PT,
                                 float* d_colpart, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (M & 255) || (N & 127) || (K & 31) || !d_A || !d_B || splits < 1) {
    hoic_set_error("hoic_mlp_gemm: M must be a multiple of 256, N of 128, K of 32"); return HOIC_ERR_ARG;
  }
  if (epi == EPI_F32 && !d_C) { hoic_set_error("hoic_mlp_gemm: float32 epilogue needs d_C"); return HOIC_ERR_ARG; }
  if (epi == EPI_FWD && !d_bias) { hoic_set_error("hoic_mlp_gemm: forward epilogue needs the bias"); return HOIC_ERR_ARG; }
  if (epi == EPI_BWD && !d_gin) { hoic_set_error("hoic_mlp_gemm: backward epilogue needs gin"); return HOIC_ERR_ARG; }
  if (epi != EPI_F32 && splits != 1) { hoic_set_error("hoic_mlp_gemm: split-K only with the float32 epilogue"); return HOIC_ERR_ARG; }
  if ((unsigned long long)M * (unsigned long long)N >= (1ull << 30)) {      // the epilogues address their outputs with 32-bit BYTE offsets
    hoic_set_error("hoic_mlp_gemm: M x N must stay below 2^30 elements"); return HOIC_ERR_ARG;
  }
  if (d_colpart && !(epi == EPI_BWD && g_gemm_pipeline == 3 && !d_PT)) {
    hoic_set_error("hoic_mlp_gemm: column partial sums come from the data-gradient epilogue in D[m][n] form (pipeline mode 3, no transposed output)");
    return HOIC_ERR_ARG;
  }
  const int nkt = K / 32;
  GemmArgs a{};
  a.A = (const u16*)d_A; a.B = (const u16*)d_B; a.M = M; a.N = N; a.K = K;
  a.kt_per_split = (nkt + splits - 1) / splits;
  a.exps = d_exps; a.amax = d_amax; a.ea = slot_a; a.eb = slot_b; a.eo = slot_out; a.extra_scale = extra_scale;
  a.C = d_C; a.c_split_stride = (long long)M * N; a.bias = d_bias; a.Gin = d_gin; a.Gout = d_gout; a.Hf32 = d_hf32;
  a.P = (u16*)d_P; a.PT = (u16*)d_PT; a.colpart = d_colpart;
  hipStream_t st = (hipStream_t)stream;
  const bool wide = (N % 256) == 0;
  if (epi == EPI_F32) return wide ? launch_gemm<256, EPI_F32>(a, splits, st) : launch_gemm<128, EPI_F32>(a, splits, st);
  if (epi == EPI_FWD) return wide ? launch_gemm<256, EPI_FWD>(a, 1, st) : launch_gemm<128, EPI_FWD>(a, 1, st);
  if (epi == EPI_BWD) return wide ? launch_gemm<256, EPI_BWD>(a, 1, st) : launch_gemm<128, EPI_BWD>(a, 1, st);
  hoic_set_error("hoic_mlp_gemm: unknown epilogue"); return HOIC_ERR_ARG;
}

extern "C" int32_t hoic_mlp_gemm_tn(int32_t M, int32_t N, int32_t K, const void* d_A, const void* d_B, const int32_t* d_exps, int32_t slot_a,
                                    int32_t slot_b, float extra_scale, int32_t splits, float* d_C, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (M & 255) || (N & 127) || (K & 31) || !d_A || !d_B || !d_C || splits < 1) {
    hoic_set_error("hoic_mlp_gemm_tn: M must be a multiple of 256, N of 128, K (rows) of 32"); return HOIC_ERR_ARG;
  }
  if ((unsigned long long)M * (unsigned long long)N >= (1ull << 32)) { hoic_set_error("hoic_mlp_gemm_tn: M x N must stay below 2^32 elements"); return HOIC_ERR_ARG; }
  GemmArgs a{};
  a.A = (const u16*)d_A; a.B = (const u16*)d_B; a.M = M; a.N = N; a.K = K;
  a.kt_per_split = (K / 32 + splits - 1) / splits;
  a.exps = d_exps; a.ea = slot_a; a.eb = slot_b; a.extra_scale = extra_scale; a.C = d_C; a.c_split_stride = (long long)M * N;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(hoic_gemm_f16x3_tn16_kernel, dim3((M / 256) * (N / 128), splits), dim3(256), 0, st, a);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

// column sums of a packed row-major tensor [R x 2C] (hi + lo), unscaled: two deterministic stages (row chunks, then their sum)
__global__ __launch_bounds__(256) void hoic_colsum_packed_kernel(const u16* __restrict__ P, int R, int Cc, int rows_per_block, float* __restrict__ part) {
  // block (x: group of 64 columns = 8 H8L8 groups, y: row chunk); thread = (column group of 8 within x: 0..7) x (row lane 0..31)
  __shared__ float red[32][65];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3, c8 = blockIdx.x * 8 + cg;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c8 * 8 < Cc)
    for (int r = r0 + rl; r < r1; r += 32) {
      const u32x4* gq = (const u32x4*)(P + ((long long)r * Cc + c8 * 8) * 2);
      const h8 h = __builtin_bit_cast(h8, gq[0]), l = __builtin_bit_cast(h8, gq[1]);
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] += (float)h[k] + (float)l[k];
    }
#pragma unroll
  for (int k = 0; k < 8; k++) red[rl][cg * 8 + k] = s[k];
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
    for (int q = 0; q < 32; q++) t += red[q][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < Cc) part[(long long)blockIdx.y * Cc + c] = t;
  }
}
__global__ void hoic_colsum_finish_kernel(const float* __restrict__ part, int nchunks, int Cc, float* __restrict__ out, const int* __restrict__ exps, int slot) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cc) return;
  float t = 0.f;
  for (int k = 0; k < nchunks; k++) t += part[(long long)k * Cc + c];
  out[c] = ldexpf(t, -(exps ? exps[slot] : 0));
}
// max |x * y| and the column sums of x * y per 128-row chunk in ONE pass over the two float32 arrays [R x C] (the last layer's
// dZ = dH * GELU'): block = 256 columns x one chunk, thread = column, rows in order => deterministic partials
__global__ __launch_bounds__(256) void hoic_amax_colsum_kernel(const float* __restrict__ x, const float* __restrict__ y, int R, int Cc, float* __restrict__ amax,
                                                               int slot, float* __restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x, r0 = blockIdx.y * 128, r1 = min(R, r0 + 128);
  float m = 0.f, s = 0.f;
  if (c < Cc) {
#pragma unroll 8
    for (int r = r0; r < r1; r++) {
      const float v = x[(long long)r * Cc + c] * y[(long long)r * Cc + c];
      m = fmaxf(m, fabsf(v)); s += v;
    }
    part[(long long)blockIdx.y * Cc + c] = s;
    if (!isfinite(s)) m = __builtin_inff();      // an Inf / NaN element survives in the sum: reported as an infinite maximum
  }
  m = wave_max_f(m);
  if ((threadIdx.x & 63) == 0 && amax) atomicMax((unsigned*)(amax + slot), __float_as_uint(m));
}
// out[c] = sum over the chunks of part[chunk][c]: 64 columns x 16 chunk phases per block, fixed order
__global__ __launch_bounds__(1024) void hoic_colpart_finish_kernel(const float* __restrict__ part, int nchunks, int Cc, float* __restrict__ out) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float t = 0.f;
  if (c < Cc) {
#pragma unroll 4
    for (int k = ph; k < nchunks; k += 16) t += part[(long long)k * Cc + c];
  }
  red[ph][cl] = t;
  __syncthreads();
  if (ph == 0 && c < Cc) {
    float u = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++) u += red[q][cl];
    out[c] = u;
  }
}
extern "C" int32_t hoic_mlp_amax_colsum(const float* d_x, const float* d_mul, int32_t R, int32_t C, float* d_amax, int32_t slot, float* d_part,
                                        void* stream) {
  if (!d_x || !d_mul || !d_part || R <= 0 || C <= 0) { hoic_set_error("hoic_mlp_amax_colsum: bad arguments"); return HOIC_ERR_ARG; }
  hipLaunchKernelGGL(hoic_amax_colsum_kernel, dim3((C + 255) / 256, (R + 127) / 128), dim3(256), 0, (hipStream_t)stream, d_x, d_mul, R, C, d_amax, slot, d_part);
  MCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int32_t hoic_mlp_colpart_finish(const float* d_part, int32_t nchunks, int32_t C, float* d_out, void* stream) {
  if (!d_part || !d_out || nchunks <= 0 || C <= 0) { hoic_set_error("hoic_mlp_colpart_finish: bad arguments"); return HOIC_ERR_ARG; }
  hipLaunchKernelGGL(hoic_colpart_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, d_part, nchunks, C, d_out);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_mlp_colsum_packed(const void* d_P, int32_t R, int32_t C, float* d_out, float* d_scratch, const int32_t* d_exps, int32_t slot,
                                          void* stream) {
  if (!d_P || !d_out || !d_scratch || R <= 0 || C <= 0 || (C & 7)) { hoic_set_error("hoic_mlp_colsum_packed: bad arguments"); return HOIC_ERR_ARG; }
  const int rows_per_block = 512, nchunks = (R + rows_per_block - 1) / rows_per_block;      // d_scratch: nchunks * C floats
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(hoic_colsum_packed_kernel, dim3((C + 63) / 64, nchunks), dim3(256), 0, st, (const u16*)d_P, R, C, rows_per_block, d_scratch);
  hipLaunchKernelGGL(hoic_colsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, d_scratch, nchunks, C, d_out, d_exps, slot);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_mlp_pack(const float* d_x, const float* d_mul, int32_t R, int32_t C, int64_t ld, void* d_P, void* d_PT, int32_t Rp,
                                 int32_t Cp, const int32_t* d_exps, int32_t slot, void* stream) {
  if (!d_x || R <= 0 || C <= 0 || Rp < R || Cp < C || (Cp & 7) || (Rp & 7)) { hoic_set_error("hoic_mlp_pack: bad arguments (padded sizes must be multiples of 8)"); return HOIC_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (d_P) {
    const long long n = (long long)Rp * (Cp >> 2);
    if (d_mul) {
      if (ld != C || Cp != C) { hoic_set_error("hoic_mlp_pack: the product form needs contiguous unpadded columns"); return HOIC_ERR_ARG; }
      hipLaunchKernelGGL(hoic_pack_rows_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_x, d_mul, R, C, (u16*)d_P, Rp, d_exps, slot);
    } else {
      hipLaunchKernelGGL(hoic_pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_x, R, C, (long long)ld, (u16*)d_P, Rp, Cp, d_exps, slot);
    }
  }
  if (d_PT) {
    if ((Rp & 63) || (Cp & 63)) { hoic_set_error("hoic_mlp_pack: the transposed form needs padded sizes that are multiples of 64"); return HOIC_ERR_ARG; }
    hipLaunchKernelGGL(hoic_pack_transpose_kernel, dim3(Cp / 64, Rp / 64), dim3(256), 0, st, d_x, d_mul, R, C, (long long)ld, (u16*)d_PT, Rp, Cp, d_exps, slot);
  }
  MCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_mlp_amax(const float* d_x, const float* d_mul, int64_t n, float* d_amax, int32_t slot, void* stream) {
  if (!d_x || n <= 0 || !d_amax) { hoic_set_error("hoic_mlp_amax: bad arguments"); return HOIC_ERR_ARG; }
  const long long want = (n + 256LL * 16 - 1) / (256LL * 16);        // ~16 elements (4 vector loads) per thread: a weight matrix takes a few hundred workgroups, not 1024
  hipLaunchKernelGGL(hoic_amax_kernel, dim3((unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want))), dim3(256), 0, (hipStream_t)stream, d_x, d_mul, (long long)n, d_amax, slot);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_mlp_update_exps(int32_t* d_exps, float* d_amax, int32_t nslots, uint64_t mask, int32_t target, int32_t exact,
                                        int32_t* d_overflow, void* stream) {
  if (!d_exps || !d_amax || nslots <= 0 || nslots > 64) { hoic_set_error("hoic_mlp_update_exps: bad arguments"); return HOIC_ERR_ARG; }
  hipLaunchKernelGGL(hoic_update_exps_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_exps, d_amax, nslots, (unsigned long long)mask, target, exact,
                     d_overflow);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

// The delayed exponents of the hidden-layer gradients, RELATIVE to the loss-side gradient whose exponent is exact: slot i gets
// target - ceil(log2 amax[i]) + (exps[ref] - *ref_prev), i.e. last pass's head-room shifted by how much the loss-side scale
// moved since -- a jump of the whole gradient's scale (new batch, clipped / unclipped ratios) cannot overflow the hidden layers.
__global__ void hoic_update_exps_rel_kernel(int* __restrict__ exps, float* __restrict__ amax, int nslots, unsigned long long mask, int target,
                                            int ref_slot, int* __restrict__ ref_prev, int* __restrict__ overflow) {
  const int i = threadIdx.x;
  const int shift = exps[ref_slot] - *ref_prev;
  __syncthreads();
  if (i < nslots && ((mask >> i) & 1ull) && i != ref_slot) {
    const float m = amax[i];
    if (m > 0.f && isfinite(m)) {
      int ex; frexpf(m, &ex);
      if (ldexpf(m, exps[i]) > 60000.f && overflow) atomicAdd(overflow, 1);
      exps[i] = target - ex + shift;
    } else if (!isfinite(m) && overflow) atomicAdd(overflow, 1);
    else exps[i] += shift;
    amax[i] = 0.f;
  }
  __syncthreads();
  if (i == 0) *ref_prev = exps[ref_slot];
}
extern "C" int32_t hoic_mlp_update_exps_rel(int32_t* d_exps, float* d_amax, int32_t nslots, uint64_t mask, int32_t target, int32_t ref_slot,
                                            int32_t* d_ref_prev, int32_t* d_overflow, void* stream) {
  if (!d_exps || !d_amax || !d_ref_prev || nslots <= 0 || nslots > 64 || ref_slot < 0 || ref_slot >= nslots) {
    hoic_set_error("hoic_mlp_update_exps_rel: bad arguments"); return HOIC_ERR_ARG;
  }
  hipLaunchKernelGGL(hoic_update_exps_rel_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_exps, d_amax, nslots, (unsigned long long)mask, target,
                     ref_slot, d_ref_prev, d_overflow);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_mlp_slab_reduce(const float* d_slabs, int32_t S, int32_t rows, int32_t cols, float* d_out, int32_t out_cols, int64_t ldo,
                                        float scale, void* stream) {
  if (!d_slabs || !d_out || S <= 0 || rows <= 0 || cols <= 0 || out_cols <= 0 || out_cols > cols) { hoic_set_error("hoic_mlp_slab_reduce: bad arguments"); return HOIC_ERR_ARG; }
  const long long n = (long long)rows * out_cols;
  hipLaunchKernelGGL(hoic_slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_slabs, S, (long long)rows * cols, rows, cols,
                     d_out, out_cols, (long long)ldo, scale);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_mlp_rowsum_packed(const void* d_P, int32_t rows, int32_t Cp, float* d_out, const int32_t* d_exps, int32_t slot, void* stream) {
  if (!d_P || !d_out || rows <= 0 || Cp <= 0 || (Cp & 7)) { hoic_set_error("hoic_mlp_rowsum_packed: bad arguments"); return HOIC_ERR_ARG; }
  hipLaunchKernelGGL(hoic_rowsum_packed_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const u16*)d_P, Cp, d_out, d_exps, slot);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

extern "C" int32_t hoic_mlp_pack_tiled(const float* d_x, int32_t R, int32_t C, int64_t ld, void* d_T, int32_t Rp, int32_t Kp, const int32_t* d_exps,
                                       int32_t slot, void* stream) {
  if (!d_x || !d_T || R <= 0 || C <= 0 || Rp < R || Kp < C || (Rp & 31) || (Kp & 15)) {
    hoic_set_error("hoic_mlp_pack_tiled: padded sizes must be multiples of 32 (rows) and 16 (columns)"); return HOIC_ERR_ARG;
  }
  const long long n = (long long)(Rp >> 5) * (Kp >> 4);
  hipLaunchKernelGGL(hoic_pack_tiled_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_x, R, C, (long long)ld, (char*)d_T, Rp, Kp, d_exps, slot);
  MCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int32_t hoic_zfilter_tiled(int32_t n, int32_t dim, const float* d_x, const double* d_state_in, double* d_state_out, int32_t update,
                                      float clip, float* d_y, double* d_scratch, void* d_T, int32_t Kp, int32_t* d_exps, int32_t slot_x,
                                      float* d_amax, int32_t nslots, uint64_t mask, int32_t target, int32_t* d_overflow, void* d_P, int32_t KpP,
                                      void* stream) {
  if (n <= 0 || dim <= 0 || !d_x || !d_state_in || !d_y || !d_T || !d_exps || (n % ZF_ROWS) || Kp < dim || (Kp & 15) ||
      nslots < 0 || nslots > 64 || (mask && !d_amax)) {
    hoic_set_error("hoic_zfilter_tiled: n must be a multiple of 128, Kp a multiple of 16 and >= dim"); return HOIC_ERR_ARG;
  }
  if (d_P && (KpP < Kp || (KpP & 7))) { hoic_set_error("hoic_zfilter_tiled: KpP must be a multiple of 8 and >= Kp"); return HOIC_ERR_ARG; }
  if (update && (!d_state_out || !d_scratch || d_state_out == d_state_in)) { hoic_set_error("hoic_zfilter_tiled: update needs a scratch buffer and a state_out that is not state_in"); return HOIC_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (update) hipLaunchKernelGGL(hoic_zfilter_moments2_kernel, dim3((unsigned)((dim + 63) / 64), (unsigned)(n / ZF_ROWS)), dim3(64), 0, st, d_x, n, dim, d_scratch);
  hipLaunchKernelGGL(hoic_zfilter_tiled_kernel, dim3((unsigned)(Kp >> 4), (unsigned)(n / 32)), dim3(64), 0, st, d_x, n, dim, d_scratch,
                     d_state_in, d_state_out, update, clip, d_y, (char*)d_T, Kp, d_exps, slot_x, d_amax, nslots, (unsigned long long)mask, target, d_overflow,
                     (u16*)d_P, KpP);
  MCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int32_t hoic_mlp_forward_tiled(int32_t M, int32_t N, int32_t K, const void* d_X, const void* d_W, const int32_t* d_exps, float* d_amax,
                                          int32_t slot_x, int32_t slot_w, int32_t slot_out, const float* d_bias, void* d_outT, float* d_outF,
                                          void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (M & 31) || (N & 63) || (K & 15) || !d_X || !d_W || !d_bias || (!d_outT == !d_outF)) {
    hoic_set_error("hoic_mlp_forward_tiled: M % 32, N % 64, K % 16 must be 0 and exactly one of d_outT / d_outF is given"); return HOIC_ERR_ARG;
  }
  FwdArgs a{};
  a.X = (const char*)d_X; a.W = (const char*)d_W; a.M = M; a.N = N; a.K = K; a.exps = d_exps; a.amax = d_amax;
  a.ex = slot_x; a.ew = slot_w; a.eo = slot_out; a.bias = d_bias; a.outT = (char*)d_outT; a.outF = d_outF;
  const dim3 grid((unsigned)((M >> 5) * (N >> 6)));
  if (d_outF) hipLaunchKernelGGL((hoic_fwd_tiled_kernel<true>), grid, dim3(64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((hoic_fwd_tiled_kernel<false>), grid, dim3(64), 0, (hipStream_t)stream, a);
  MCHK(hipGetLastError());
  return HOIC_OK;
}


// ---------------------------------------------------------------------------------------------- the action head of the rollout
// out[m][n] = sum_k h[m][k] W[n][k] + bias[n] (+ std[n] * eps[m][n]):  PolicyGaussian.action_mean on the MLP body's output and
// the Gaussian sample (uhc/khrylib/rl/core/policy_gaussian.py:27-33, distributions.py: mean + std * N(0, 1)) in ONE launch
// that needs no LDS, so that it runs beside the simulator's substep workgroups like hoic_fwd_tiled_kernel does (the library
// GEMM of this 2048 x 512 x 32 product queues for a CU's LDS behind them, and the sample was two more launches).
// float32 throughout: the f32 MFMA is an exact float32 multiply-add chain.
__global__ __launch_bounds__(64) void hoic_head_kernel(int M, int K, int N, const float* __restrict__ h, long long ldh, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const float* __restrict__ stdv, const float* __restrict__ eps,
                                                        long long lde, float* __restrict__ out, long long ldo) {
  // One wavefront = 16 rows x 32 outputs as two v_mfma_f32_16x16x4_f32 tiles (two independent accumulators: the MFMA chain of
  // one tile runs in the shadow of the other's).  Lane (r = lane & 15, g = lane >> 4) loads four consecutive k of row r (h)
  // and of outputs n = r, 16 + r (W) per 16-k block -- group g the g-th four -- and MFMA step s contracts element s of all four
  // groups, so every operand is a 16-byte load and both matrices are read as stored.
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4, row0 = blockIdx.x * 16;
  const float* hp = h + (long long)min(row0 + r, M - 1) * ldh + 4 * g;       // rows past M are computed (from row M - 1) and not written
  const float* w0 = W + (long long)(r < N ? r : 0) * K + 4 * g;
  const float* w1 = W + (long long)(16 + r < N ? 16 + r : 0) * K + 4 * g;
  const float m0 = r < N ? 1.f : 0.f, m1 = 16 + r < N ? 1.f : 0.f;
  f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int kb = 0; kb < K; kb += 16) {
    const f32x4 a = *(const f32x4*)(hp + kb);
    const f32x4 b0 = *(const f32x4*)(w0 + kb) * m0, b1 = *(const f32x4*)(w1 + kb) * m1;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc1, 0, 0, 0);
    }
  }
  // D[i][j] of a 16 x 16 tile: lane (j + 16 g) holds rows i = 4 g + reg of output column j
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const int n = 16 * t + r;
    if (n >= N) continue;
    const f32x4v& acc = t ? acc1 : acc0;
    const float bj = bias ? bias[n] : 0.f, sj = (eps && stdv) ? stdv[n] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int i = row0 + 4 * g + reg;
      if (i < M) {
        float v = acc[reg] + bj;
        if (eps) v = fmaf(sj, eps[(long long)i * lde + n], v);
        out[(long long)i * ldo + n] = v;
      }
    }
  }
}
extern "C" int32_t hoic_mlp_head(int32_t M, int32_t K, int32_t N, const float* d_h, int64_t ldh, const float* d_W, const float* d_bias,
                                 const float* d_std, const float* d_eps, int64_t lde, float* d_out, int64_t ldo, void* stream) {
  if (M <= 0 || K <= 0 || (K & 15) || N <= 0 || N > 32 || !d_h || !d_W || !d_out || (ldh & 3) || ((size_t)d_h & 15) || ((size_t)d_W & 15)) {
    hoic_set_error("hoic_mlp_head: K % 16 == 0, N <= 32, 16-byte aligned h / W with ldh % 4 == 0"); return HOIC_ERR_ARG;
  }
  hipLaunchKernelGGL(hoic_head_kernel, dim3((unsigned)((M + 15) >> 4)), dim3(64), 0, (hipStream_t)stream, M, K, N, d_h, (long long)ldh, d_W, d_bias, d_std, d_eps,
                     (long long)lde, d_out, (long long)ldo);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

// ---------------------------------------------------------------------------------------------- the heads' backward pass
// The head of either network is out = h W^T + b with N <= 32 outputs on the K-wide output of the MLP body (action_mean of
// policy_gaussian.py:16-25, value_head of the value MLP).  Given g = dLoss/dout [M, N] this launch makes, in ONE pass over h,
//   dh[m][k] = sum_n g[m][n] W[n][k]                       (what the body's backward pass starts from)
//   part[blk][n][k] = sum_{m in blk} g[m][n] h[m][k]       (weight gradient, finished by hoic_colpart_finish_kernel in fixed order)
//   part[blk][N K + n] = sum_{m in blk} g[m][n]            (bias gradient)
// in float32 FMAs.  A thread owns two adjacent columns k: the W entries of its columns and its 2 N partial sums stay in
// registers, g[m][:] is one row of <= 32 floats every thread of the block reads from the same address (a broadcast), h and
// dh move as coalesced 8-byte accesses -- the launch is bound by reading h and writing dh once (2 x 4 K bytes per row).
// It replaces three library GEMMs per head and step; those were the only library GEMMs inside the update's two concurrent
// chains, and the library's stream-K kernel for the weight gradient is not safe to run on two streams at once.
#define HEADB_NT 256
template <int NMAX>
__global__ __launch_bounds__(HEADB_NT) void hoic_head_bwd_kernel(int M, int K, int N, int rows_per_block, const float* __restrict__ h, long long ldh,
                                                                 const float* __restrict__ W, const float* __restrict__ g, long long ldg,
                                                                 float* __restrict__ dh, long long lddh, float* __restrict__ part) {
  const int tid = threadIdx.x, m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float* mypart = part + (long long)blockIdx.x * (((long long)N * K + N + 1) & ~1LL);      // even stride: 8-byte stores
  for (int k0 = 2 * tid; k0 < K; k0 += 2 * HEADB_NT) {
    float w0[NMAX], w1[NMAX], a0[NMAX], a1[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; n++) {
      const float2 w = n < N ? *(const float2*)(W + (long long)n * K + k0) : make_float2(0.f, 0.f);
      w0[n] = w.x; w1[n] = w.y; a0[n] = 0.f; a1[n] = 0.f;
    }
    for (int m = m0; m < m1; m++) {
      const float2 hv = *(const float2*)(h + (long long)m * ldh + k0);
      const float* gr = g + (long long)m * ldg;
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int n = 0; n < NMAX; n++) {
        const float gn = n < N ? gr[n] : 0.f;
        d0 = fmaf(gn, w0[n], d0); d1 = fmaf(gn, w1[n], d1);
        a0[n] = fmaf(gn, hv.x, a0[n]); a1[n] = fmaf(gn, hv.y, a1[n]);
      }
      *(float2*)(dh + (long long)m * lddh + k0) = make_float2(d0, d1);
    }
#pragma unroll
    for (int n = 0; n < NMAX; n++)
      if (n < N) *(float2*)(mypart + (long long)n * K + k0) = make_float2(a0[n], a1[n]);
  }
  if (tid < N) {
    float sb = 0.f;
    for (int m = m0; m < m1; m++) sb += g[(long long)m * ldg + tid];
    mypart[(long long)N * K + tid] = sb;
  }
}
// The same launch for 8 < N <= 32 outputs (the action head) on the matrix core: per row the two products are 2 x 32 x 512
// multiply-adds -- 227 us per launch as FMAs, where reading h and writing dh once take ~55.  A block of eight wavefronts walks
// its rows in tiles of 32; wavefront w owns columns [64 w, 64 w + 64) as two 32-column blocks.  Per tile and block, as
// v_mfma_f32_32x32x2_f32 (A: lane i + 32 s holds A[i][s], B: B[s][j], D[i][j]: lane j + 32 s holds rows 8 q + 4 s + r):
//   dh tile [32 rows x 32 cols] = sum_t  g[row i][n = 2 t + s]  x  W[n = 2 t + s][col j]        (16 steps; W in 32 registers)
//   dW block [32 n x 32 cols]  += sum_t  g[row 2 t + s][n = i]  x  h[row 2 t + s][col j]        (16 steps; 2 x 16 accumulators)
// Both operand layouts of the g tile come from one 32 x 33 float LDS image; h, dh and the partial sums move as 128-byte row
// segments.  The f32 MFMA is an exact float32 multiply-add chain, so this is the arithmetic of the FMA form in another order.
#define HEADM_NC 2          // 32-column blocks per wavefront
#define HEADM_NT 512        // 8 wavefronts x 64 columns = 512 columns per pass
__global__ __launch_bounds__(HEADM_NT) void hoic_head_bwd_mfma_kernel(int M, int K, int N, int rows_per_block, const float* __restrict__ h, long long ldh,
                                                                    const float* __restrict__ W, const float* __restrict__ g, long long ldg,
                                                                    float* __restrict__ dh, long long lddh, float* __restrict__ part) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  __shared__ float gl[32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, s = lane >> 5;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float* mypart = part + (long long)blockIdx.x * (((long long)N * K + N + 1) & ~1LL);
  for (int kbase = 0; kbase < K; kbase += 512) {
    const int kc = kbase + wave * 32 * HEADM_NC;
    float wreg[HEADM_NC][16];
    f32x16 acc[HEADM_NC];
#pragma unroll
    for (int c = 0; c < HEADM_NC; c++) {
      const int col = min(kc + 32 * c + l32, K - 1);
#pragma unroll
      for (int t = 0; t < 16; t++) wreg[c][t] = 2 * t + s < N ? W[(long long)(2 * t + s) * K + col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    }
    float bsum = 0.f;        // bias gradient: wavefront 0, lane n (first pass over the columns only)
    for (int mt = m0; mt < m1; mt += 32) {
      __syncthreads();
      for (int e = threadIdx.x; e < 1024; e += HEADM_NT) {
        const int r = e >> 5, c = e & 31, row = mt + r;
        gl[r][c] = (row < m1 && c < N) ? g[(long long)row * ldg + c] : 0.f;
      }
      __syncthreads();
      float ga[16], gb[16];
#pragma unroll
      for (int t = 0; t < 16; t++) { ga[t] = gl[l32][2 * t + s]; gb[t] = gl[2 * t + s][l32]; }
      if (kbase == 0 && wave == 0 && s == 0) {
#pragma unroll
        for (int r = 0; r < 32; r++) bsum += gl[r][l32];
      }
#pragma unroll
      for (int c = 0; c < HEADM_NC; c++) {
        if (kc + 32 * c >= K) continue;
        const int col = kc + 32 * c + l32;
        float hb[16];
#pragma unroll
        for (int t = 0; t < 16; t++) hb[t] = h[(long long)min(mt + 2 * t + s, M - 1) * ldh + col];
        f32x16 d;
#pragma unroll
        for (int r = 0; r < 16; r++) d[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; t++) {
          d = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[t], wreg[c][t], d, 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(gb[t], hb[t], acc[c], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = mt + 8 * (r >> 2) + 4 * s + (r & 3);
          if (row < m1) dh[(long long)row * lddh + col] = d[r];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < HEADM_NC; c++) {
      if (kc + 32 * c >= K) continue;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int n = 8 * (r >> 2) + 4 * s + (r & 3);
        if (n < N) mypart[(long long)n * K + kc + 32 * c + l32] = acc[c][r];
      }
    }
    if (kbase == 0 && wave == 0 && s == 0 && l32 < N) mypart[(long long)N * K + l32] = bsum;
  }
}
extern "C" int32_t hoic_mlp_head_backward(int32_t M, int32_t K, int32_t N, const float* d_h, int64_t ldh, const float* d_W, const float* d_g, int64_t ldg,
                                          float* d_dh, int64_t lddh, float* d_grad, float* d_part, int32_t nblocks, void* stream) {
  if (M <= 0 || K <= 0 || (K & 1) || N <= 0 || N > 32 || nblocks <= 0 || !d_h || !d_W || !d_g || !d_dh || !d_grad || !d_part || (ldh & 1) || (lddh & 1) ||
      ((size_t)d_h & 7) || ((size_t)d_W & 7) || ((size_t)d_dh & 7) || ((size_t)d_part & 7)) {
    hoic_set_error("hoic_mlp_head_backward: K even, N <= 32, 8-byte aligned h / W / dh / part with even leading dimensions"); return HOIC_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const int rows = (M + nblocks - 1) / nblocks;
  int nb = (M + rows - 1) / rows;       // nb <= nblocks blocks, the last one ragged
  if (N > 8 && (K & 31) == 0) {       // matrix-core form: rows in tiles of 32
    const int tiles = (M + 31) / 32, nbm = std::min(nblocks, 256), tpb = (tiles + nbm - 1) / nbm;      // one 8-wavefront block per CU
    nb = (tiles + tpb - 1) / tpb;
    hipLaunchKernelGGL(hoic_head_bwd_mfma_kernel, dim3(nb), dim3(HEADM_NT), 0, st, M, K, N, 32 * tpb, d_h, (long long)ldh, d_W, d_g, (long long)ldg, d_dh, (long long)lddh, d_part);
  } else if (N <= 8)
    hipLaunchKernelGGL(hoic_head_bwd_kernel<8>, dim3(nb), dim3(HEADB_NT), 0, st, M, K, N, rows, d_h, (long long)ldh, d_W, d_g, (long long)ldg, d_dh, (long long)lddh, d_part);
  else
    hipLaunchKernelGGL(hoic_head_bwd_kernel<32>, dim3(nb), dim3(HEADB_NT), 0, st, M, K, N, rows, d_h, (long long)ldh, d_W, d_g, (long long)ldg, d_dh, (long long)lddh, d_part);
  const int C = (N * K + N + 1) & ~1;       // the stride of a block's partial sums
  hipLaunchKernelGGL(hoic_colpart_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, d_part, nb, C, d_grad);
  MCHK(hipGetLastError());
  return HOIC_OK;
}

// ---------------------------------------------------------------------------------------------- the two losses of the update
// hoic_ppo_loss_kernel: everything between the action head's output and its gradient in ONE launch --
//   log pi(a|s) = sum_j [ -(a_j - mean_j)^2 / (2 exp(2 s_j)) - s_j - log sqrt(2 pi) ]   (policy_gaussian.py / distributions.py log_prob, s = action_log_std)
//   ratio = exp(log pi - fixed),  surr1 = ratio A,  surr2 = clamp(ratio, 1 - eps, 1 + eps) A,  L = -mean(min(surr1, surr2))   (agent_ppo.py:58-64)
// and the backward pass torch.autograd makes of it: dL/dmean -> g, the sums over the rows for dL/ds_j and L itself as per-block
// partial sums (finished in fixed order by hoic_colpart_finish_kernel).  fixed == NULL is epoch 0 of agent_ppo.py:18-20: the
// old policy IS the current one, ratio = exp(0) = 1 exactly, and the log-probabilities go to logp_out for the later epochs.
// A row is a 32-lane half wavefront (lane = action dimension): 128-byte row accesses, the sum over j by 5 shuffles.
// The gradient of min / clamp follows PyTorch's rules: minimum passes the gradient to the smaller argument (half to each at a
// tie), clamp passes it inside the closed interval -- together: pass iff surr1 < surr2 or ratio in [1 - eps, 1 + eps].
#define LOSS_PART 34         // per-block partial sums: 32 x dL/ds_j, L, one pad
__global__ __launch_bounds__(256) void hoic_ppo_loss_kernel(int M, int N, const float* __restrict__ mean, long long ldm, const float* __restrict__ act, long long lda,
                                                            const float* __restrict__ adv, const float* __restrict__ fixed, const float* __restrict__ log_std,
                                                            float clip, float scale, float inv_m, float* __restrict__ g, long long ldg,
                                                            float* __restrict__ logp_out, float* __restrict__ part) {
  __shared__ float red[8][LOSS_PART];
  const int j = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const bool on = j < N;
  const float ls = on ? log_std[j] : 0.f, var = expf(2.f * ls);
  float acc_ls = 0.f, acc_loss = 0.f;
  for (long long m = (long long)blockIdx.x * 8 + hw; m < M; m += (long long)gridDim.x * 8) {
    const float d = on ? act[m * lda + j] - mean[m * ldm + j] : 0.f;
    float lp = on ? -(d * d) / (2.f * var) - ls - 0.91893853320467274f : 0.f;
#pragma unroll
    for (int o = 16; o; o >>= 1) lp += __shfl_xor(lp, o, 32);
    const float r = fixed ? expf(lp - fixed[m]) : 1.f, A = adv[m];
    const float s1 = r * A, s2 = fminf(fmaxf(r, 1.f - clip), 1.f + clip) * A;
    const bool pass = s1 < s2 || (r >= 1.f - clip && r <= 1.f + clip);
    const float dlp = pass ? -scale * A * r : 0.f;
    if (on) {
      g[m * ldg + j] = dlp * d / var;
      acc_ls = fmaf(dlp, d * d / var - 1.f, acc_ls);
    }
    if (j == 0) { acc_loss -= fminf(s1, s2) * inv_m; if (logp_out) logp_out[m] = lp; }
  }
  red[hw][j] = acc_ls;
  if (j == 0) { red[hw][32] = acc_loss; red[hw][33] = 0.f; }
  __syncthreads();
  if (threadIdx.x < LOSS_PART) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) t += red[q][threadIdx.x];
    part[(long long)blockIdx.x * LOSS_PART + threadIdx.x] = t;
  }
}
// hoic_value_loss_kernel: L = mean((v - ret)^2) (agent_pg.py:18-25 update_value), g = dL/dv = 2 (v - ret) weight / M
__global__ __launch_bounds__(256) void hoic_value_loss_kernel(int M, const float* __restrict__ v, const float* __restrict__ ret, float scale2, float inv_m,
                                                              float* __restrict__ g, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    const float e = v[m] - ret[m];
    g[m] = scale2 * e;
    acc = fmaf(e * inv_m, e, acc);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
extern "C" int32_t hoic_mlp_ppo_loss(int32_t M, int32_t N, const float* d_mean, int64_t ldm, const float* d_act, int64_t lda, const float* d_adv,
                                     const float* d_fixed, const float* d_log_std, float clip, float weight, float* d_g, int64_t ldg, float* d_logp_out,
                                     float* d_sums, float* d_part, int32_t nblocks, void* stream) {
  if (M <= 0 || N <= 0 || N > 32 || nblocks <= 0 || !d_mean || !d_act || !d_adv || !d_log_std || !d_g || !d_sums || !d_part || (!d_fixed && !d_logp_out)) {
    hoic_set_error("hoic_mlp_ppo_loss: N <= 32, and without fixed log-probabilities (epoch 0) a place to put them"); return HOIC_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)std::min<long long>(nblocks, ((long long)M + 7) / 8);
  hipLaunchKernelGGL(hoic_ppo_loss_kernel, dim3(nb), dim3(256), 0, st, M, N, d_mean, (long long)ldm, d_act, (long long)lda, d_adv, d_fixed, d_log_std, clip,
                     weight / (float)M, 1.f / (float)M, d_g, (long long)ldg, d_logp_out, d_part);
  hipLaunchKernelGGL(hoic_colpart_finish_kernel, dim3(1), dim3(1024), 0, st, d_part, nb, LOSS_PART, d_sums);
  MCHK(hipGetLastError());
  return HOIC_OK;
}
extern "C" int32_t hoic_mlp_value_loss(int32_t M, const float* d_v, const float* d_ret, float weight, float* d_g, float* d_loss, float* d_part,
                                       int32_t nblocks, void* stream) {
  if (M <= 0 || nblocks <= 0 || !d_v || !d_ret || !d_g || !d_loss || !d_part) { hoic_set_error("hoic_mlp_value_loss: bad arguments"); return HOIC_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)std::min<long long>(nblocks, ((long long)M + 255) / 256);
  hipLaunchKernelGGL(hoic_value_loss_kernel, dim3(nb), dim3(256), 0, st, M, d_v, d_ret, 2.f * weight / (float)M, 1.f / (float)M, d_g, d_part);
  hipLaunchKernelGGL(hoic_colpart_finish_kernel, dim3(1), dim3(1024), 0, st, d_part, nb, 1, d_loss);
  MCHK(hipGetLastError());
  return HOIC_OK;
}
