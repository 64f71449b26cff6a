// Main radiance field (ThermalNerfactoField): 16-level hash grid + fused fp32-MFMA MLP chain.
//   Field.forward = NerfactoField.get_density + get_outputs   (fields/base_field.py:114-133, fields/nerfacto_field.py:205-229,272-348,
//                                                              fields/thermal_nerfacto_field.py:91-99)
//
// Kernels:
//   k_field_pos        streaming    : contracted position + selector of every sample, once
//   k_field_encode_xcd gather       : XCD-affine and level-major -- block b runs on XCD b % 8 and gathers from ONE level, so that a hashed
//                                     level's 4 MB table is served from that XCD's L2 while all samples visit it (thread = 4 samples)
//   k_field_mlp_fwd    MFMA chain   : one wave = 32 samples: Linear(32,64) ReLU Linear(64,16) | head Linear(64 slots,64) ReLU Linear(64,64)
//                                     ReLU Linear(64,C) sigmoid, every layer as v_mfma_f32_32x32x2_f32 on OUT^T = W . IN^T so that a layer's
//                                     accumulator tile IS the next layer's B operand (no LDS round trip, no cross-lane traffic between layers).
//   k_field_bwd_fused  MFMA chain   : dIN^T = W^T . dOUT^T with the same trick AND every weight gradient dW = dY^T X (operands transposed
//                                     through per-wave LDS tiles), bias sums, appearance-embedding rows: one launch, one wave per SIMD
//   k_field_dpos       streaming    : d position from d enc and the forward's saved d enc / d offset, beside the table scatter
//   k_field_density_only            : the first two layers alone: the density-only evaluation (config 2's cross terms)
//   table-gradient scatter: tn_scatter.hip
//
// Register layout used everywhere ("D-layout" of v_mfma_f32_32x32x2_f32): lane = (j = lane&31 -> sample in the tile,
// h = lane>>5), accumulator register r in [0,16) holds feature row  R(r,h) = (r&3) + 8*(r>>2) + 4*h  of a 32-row tile.
// Exact fp32: the MFMA is a k-ordered fmaf chain (no reduced precision), which the 1e-4 density tolerance needs.
#include "tn_common.h"
#include "tn_field_dpos.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
#ifndef FRAG_NT
#define FRAG_NT 1  // non-temporal stores / loads of the saved activations (see store_frag)
#endif

#define TILE 32
#define FRAG 64  // floats per A fragment (one per lane)

__host__ __device__ __forceinline__ constexpr int RROW(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- packed-weight layout inside the workspace (float offsets) -------------------------------------------------------
// forward fragments  Af[layer][m][t][r][lane] = W[32m + (lane&31)][32t + R(r, lane>>5)]
// backward fragments Ab[layer][t][m][r][lane] = W[32m + R(r, lane>>5)][32t + (lane&31)]      (output tile t = input features)
// layers: 0 base0 (out 64, in 32)  1 base1 (out 16, in 64)  2 head0 (out 64, in 64 slots)  3 head1 (64,64)  4 head2 (out C, in 64)
struct LayerGeom { int mo, ti; };  // output tiles, input tiles (of 32)
__host__ __device__ __forceinline__ constexpr int layer_mo(int l) { return (l == 0 || l == 2 || l == 3) ? 2 : 1; }
__host__ __device__ __forceinline__ constexpr int layer_ti(int l) { return l == 0 ? 1 : 2; }
__host__ __device__ __forceinline__ constexpr int layer_frags(int l) { return layer_mo(l) * layer_ti(l) * 16; }
__host__ __device__ __forceinline__ constexpr int fwd_off(int l) {
  int o = 0;
  for (int i = 0; i < l; ++i) o += layer_frags(i) * FRAG;
  return o;
}
#define PACK_FWD_FLOATS (fwd_off(5))                      // 14336
#define PACK_BIAS_OFF PACK_FWD_FLOATS                     // biases: 5 layers x 64 floats (padded)
#define PACK_BIAS_FLOATS (5 * 64)
#define PACK_MEANEMB_OFF (PACK_BIAS_OFF + PACK_BIAS_FLOATS)  // 32 floats: mean appearance embedding
#define PACK_FWD_TOTAL (PACK_MEANEMB_OFF + 32)               // what the forward kernel stages in LDS (14688 floats = 58,752 B)
#define PACK_BWD_OFF 14720                                   // (PACK_FWD_TOTAL rounded up to 64)
#define PACK_BWD_FLOATS PACK_FWD_FLOATS
#define PACK_TOTAL_FLOATS (PACK_BWD_OFF + PACK_BWD_FLOATS)   // 29056 floats
static_assert(PACK_FWD_TOTAL <= PACK_BWD_OFF, "pack layout");
// Behind the fragments, in the same workspace region: the appearance embedding's share of the colour head's first layer, per camera --
//   camhead[cam][f] = sum_e hw0[f][31 + e] * emb[cam][e]
// The embedding is constant over a camera's samples, so the TRAINING forward starts head layer 0 from bias + camhead[cam] instead of running
// the 32 embedding slots of every sample through the matrix cores (32 of the layer's 64 MFMAs per tile).  Rebuilt with the fragments.
#define FIELD_MAX_IMAGES 4096  // cameras the per-camera tables are sized for (camhead here, the backward's per-camera sums: 1 MB each)
#define PACK_CAMHEAD_OFF PACK_TOTAL_FLOATS
// Behind camhead: the colour head's two 64-wide layers (2 = head0, 3 = head1) as SPLIT-bf16 fragments for v_mfma_f32_32x32x16_bf16 (the opt-in
// TN_HEAD_BF16X3=1 path: w = hi + lo with hi = bf16(w), lo = bf16(w - hi); a product is hi*hi + hi*lo + lo*hi in fp32 accumulators, ~2^-16
// relative).  A word = two bf16; a fragment = 64 lanes x 4 words (one ds_read_b128 per lane):
//   forward  Bf[layer-2][m][t][s][part][lane][q] = part(W[32m + (lane&31)][32t + R(8s + q, lane>>5)])      q = 0..7, part 0 = hi, 1 = lo
//   backward Bb[layer-2][t][m][s][part][lane][q] = part(W[32m + R(8s + q, lane>>5)][32t + (lane&31)])
// -- k-step s of a 32-feature tile covers the D-layout registers 8s..8s+7 of both lane halves (k = 8h + q <-> row R(8s + q, h)), so an fp32
// accumulator tile, split on the vector ALU, is the next layer's B operand exactly as on the fp32 path.
#define PACK_BF_OFF (PACK_TOTAL_FLOATS + FIELD_MAX_IMAGES * 64)
#define PACK_BF_LAYER_WORDS (2 * 2 * 2 * 2 * 256)  // [m][t][s][part] fragments of 256 words
#define PACK_BF_FWD_WORDS (2 * PACK_BF_LAYER_WORDS)
#define PACK_BF_BWD_OFF (PACK_BF_OFF + PACK_BF_FWD_WORDS)
#define PACK_BF_WORDS (2 * PACK_BF_FWD_WORDS)
#define PACK_REGION_FLOATS (PACK_BF_OFF + PACK_BF_WORDS)
static_assert(PACK_CAMHEAD_OFF % 64 == 0 && PACK_BF_OFF % 64 == 0, "pack layout");

// slot space of the colour head's first layer: [0,16) SH, [16,32) base-MLP output rows 0..15 (row 0 = density logit, weight 0),
// [32,64) appearance embedding.  nn.Linear column for a slot (or -1):
__host__ __device__ __forceinline__ constexpr int slot_to_col(int s) { return s < 16 ? s : (s == 16 ? -1 : s - 1); }

struct FieldK {
  const float *w0, *b0, *w1, *b1, *hw0, *hb0, *hw1, *hb1, *hw2, *hb2, *emb;
  int C, num_images;
};

__device__ __forceinline__ float weight_at(const FieldK& f, int layer, int o, int i) {
  switch (layer) {
    case 0: return (o < 64 && i < 32) ? f.w0[o * 32 + i] : 0.0f;
    case 1: return (o < 16 && i < 64) ? f.w1[o * 64 + i] : 0.0f;
    case 2: { int c = slot_to_col(i); return (o < 64 && i < 64 && c >= 0) ? f.hw0[o * 63 + c] : 0.0f; }
    case 3: return (o < 64 && i < 64) ? f.hw1[o * 64 + i] : 0.0f;
    default: return (o < f.C && i < 64) ? f.hw2[o * 64 + i] : 0.0f;
  }
}
__device__ __forceinline__ float bias_at(const FieldK& f, int layer, int o) {
  switch (layer) {
    case 0: return o < 64 ? f.b0[o] : 0.0f;
    case 1: return o < 16 ? f.b1[o] : 0.0f;
    case 2: return o < 64 ? f.hb0[o] : 0.0f;
    case 3: return o < 64 ? f.hb1[o] : 0.0f;
    default: return o < f.C ? f.hb2[o] : 0.0f;
  }
}

#define PACK_BLOCKS ((PACK_FWD_TOTAL + 255) / 256)
__device__ __forceinline__ void field_pack_body(const FieldK& f, float* __restrict__ pack, int idx) {
  if (idx < PACK_FWD_FLOATS) {
    int layer = 0;
    while (layer < 4 && idx >= fwd_off(layer + 1)) ++layer;
    int rel = idx - fwd_off(layer);
    int lane = rel & 63, frag = rel >> 6;
    int r = frag & 15, mt = frag >> 4;
    int ti = layer_ti(layer);
    // forward: frag order [m][t][r]
    int m = mt / ti, t = mt % ti;
    pack[idx] = weight_at(f, layer, 32 * m + (lane & 31), 32 * t + RROW(r, lane >> 5));
    // backward: frag order [t][m][r] (same count)
    int mo = layer_mo(layer);
    int tb = mt / mo, mb = mt % mo;
    pack[PACK_BWD_OFF + idx] = weight_at(f, layer, 32 * mb + RROW(r, lane >> 5), 32 * tb + (lane & 31));
  } else if (idx < PACK_FWD_FLOATS + PACK_BIAS_FLOATS) {
    int q = idx - PACK_FWD_FLOATS;
    pack[idx] = bias_at(f, q >> 6, q & 63);
  } else if (idx < PACK_FWD_TOTAL) {
    int e = idx - PACK_MEANEMB_OFF;
    // Embedding.mean(dim=0) (field_components/embedding.py:45-47)
    float s = 0.0f;
    for (int c = 0; c < f.num_images; ++c) s += f.emb[c * 32 + e];
    pack[idx] = s / (float)f.num_images;
  } else if (idx >= PACK_BLOCKS * 256) {  // camhead: one thread per (camera, hidden unit)
    const int q = idx - PACK_BLOCKS * 256;
    const int ncam = f.num_images < FIELD_MAX_IMAGES ? f.num_images : FIELD_MAX_IMAGES;
    if (q < ncam * 64) {
      const float* w = f.hw0 + (q & 63) * 63 + 31;
      const float* e = f.emb + (q >> 6) * 32;
      float s = 0.0f;
      for (int k = 0; k < 32; ++k) s = fmaf(w[k], e[k], s);
      pack[PACK_CAMHEAD_OFF + q] = s;
    }
    const int w = q - ((ncam * 64 + 255) / 256) * 256;  // split-bf16 fragments of the colour head: one thread per word (two weights)
    if (w >= 0 && w < PACK_BF_WORDS) {
      const bool bwd = w >= PACK_BF_FWD_WORDS;
      const int rel = bwd ? w - PACK_BF_FWD_WORDS : w;
      const int v = rel & 3, lane = (rel >> 2) & 63, part = (rel >> 8) & 1, s8 = (rel >> 9) & 1, i1 = (rel >> 10) & 1, i0 = (rel >> 11) & 1;
      const int layer = 2 + (rel >> 12);
      uint32_t word = 0;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kk = RROW(8 * s8 + 2 * v + e, lane >> 5);
        // forward [m = i0][t = i1]: W[32m + lane][32t + kk]; backward [t = i0][m = i1]: W[32m + kk][32t + lane]
        const float x = bwd ? weight_at(f, layer, 32 * i1 + kk, 32 * i0 + (lane & 31)) : weight_at(f, layer, 32 * i0 + (lane & 31), 32 * i1 + kk);
        const __bf16 hi = (__bf16)x;
        const __bf16 val = part ? (__bf16)(x - (float)hi) : hi;
        word |= (uint32_t)__builtin_bit_cast(unsigned short, val) << (16 * e);
      }
      reinterpret_cast<uint32_t*>(pack)[PACK_BF_OFF + w] = word;
    }
  }
}
__global__ void k_field_pack(FieldK f, float* __restrict__ pack) { field_pack_body(f, pack, blockIdx.x * blockDim.x + threadIdx.x); }
static inline int pack_blocks(int num_images) {  // fragment blocks + one thread per (camera, hidden unit) of camhead + the split-bf16 words
  return PACK_BLOCKS + ((num_images < FIELD_MAX_IMAGES ? num_images : FIELD_MAX_IMAGES) * 64 + 255) / 256 + PACK_BF_WORDS / 256;
}

// ---- workspace layout (byte offsets; every region 256-B aligned) ------------------------------------------------------
// Per-level tensors (enc, d enc / d offset, d enc) are LEVEL-MAJOR with a row stride of PT = samples rounded up to whole 32-sample tiles:
// the XCD-affine encode writes one level at a time, the MFMA kernels read 8 levels per lane as coalesced float2 runs.
// Saved activations are kept in FRAGMENT ORDER: [tile of 32 samples][m (32-feature tile)][g][lane] float4, where lane (j = sample, h) of
// register group g holds features 32 m + 8 g + 4 h + {0..3} -- exactly what a wave holds in the D-layout, so every store / load instruction
// of a wave moves one contiguous KiB (sample-major rows cost four 32-byte pieces per 128-byte line: 4x the L2 transactions).
// Nothing outside this file reads them.
struct FieldWs {
  float* pack;     // PACK_REGION_FLOATS: forward / backward fragments, biases, mean embedding, then camhead
  float* pos;      // [P] float4: contracted position in [0,1]^3 (masked) and the selector (k_field_pos)
  float* enc;      // [16][PT] float2   hash encoding, level-major
  float* sel;      // [P]
  float* sh;       // [rays][2][8]: SH16 of the ray's direction in D-layout order (entry [h][r] = sh[R(r,h)]); sized for S = 1
  // training only
  float* h1;       // [tiles][2][4][64] float4   relu(base layer 0)
  float* hin;      // [tiles][2][64] float4      the 16 base-MLP outputs (head-input slots 16..31; SH and the appearance embedding are per-ray:
                   //                            the backward rebuilds them)
  float* hh1;      // relu(head layer 0)
  float* hh2;      // relu(head layer 1)
  float* y;        // [P][4]   sigmoid outputs (for the sigmoid derivative)
  float* g_enc;    // [16][P] float2  d enc, LEVEL-major (read by the table scatter's bin pass and k_field_dpos)
  float* jac;      // res * d enc / d offset: [16 levels][3 axes][PT] float2 = (feature 0, feature 1) (written by the training encode, read by
                   // k_field_dpos)
  float* cam_bias; // [FIELD_MAX_IMAGES][64]  head layer 0's bias gradient per camera (k_field_bwd_fused -> k_field_emb_finish); zero between uses
  void* scatter;   // scratch of the table-gradient scatter (tn_scatter_scratch_bytes)
  int64_t PT;      // row stride of the level-major tensors
  int64_t bytes;
};
static inline FieldWs ws_layout(void* base, int64_t P, int training) {
  FieldWs w;
  char* p = reinterpret_cast<char*>(base);
  int64_t off = 0;
  auto take = [&](int64_t floats) {
    float* r = reinterpret_cast<float*>(p + off);
    off += ((floats * 4 + 255) / 256) * 256;
    return r;
  };
  const int64_t PT = tn_cdiv(P, 32) * 32;  // whole tiles
  w.PT = PT;
  w.pack = take(PACK_REGION_FLOATS);
  w.pos = take(P * 4);
  w.enc = take(PT * 32);
  w.sel = take(P);
  w.sh = take(P * 16);
  if (training) {
    w.h1 = take(PT * 64); w.hin = take(PT * 16); w.hh1 = take(PT * 64); w.hh2 = take(PT * 64); w.y = take(P * 4);
    w.g_enc = take(P * 32);
    w.jac = take(PT * 96);
    w.cam_bias = take((int64_t)FIELD_MAX_IMAGES * 64 + 64);  // (+ the finishing launch's block counter, one 256-B line behind the sums)
    w.scatter = take(tn_scatter_scratch_bytes(P, TN_MAX_LEVELS) / 4);
  } else {
    w.h1 = w.hin = w.hh1 = w.hh2 = w.y = w.g_enc = w.jac = w.cam_bias = nullptr;
    w.scatter = nullptr;
  }
  w.bytes = off;
  return w;
}
extern "C" int64_t tn_field_workspace_bytes(int64_t num_points, int32_t training) {
  if (num_points < 0) return TN_EINVAL;
  return ws_layout(nullptr, num_points, training).bytes;
}

// ---- positions ---------------------------------------------------------------------------------------------------------
// thread = sample: Frustums.get_positions + SceneContraction + (x+2)/4 + selector, once per sample (the encode below visits a sample once per
// LEVEL and reads the 16 bytes written here instead of repeating two loads of the ray, the division by S and the contraction)
// PACK: the first PACK_BLOCKS blocks of the same launch rebuild the packed weights (tn_field_pack_weights: the parameters changed since the last
// step) and every other block first clears `zero` (the step's accumulators: loss lines, d comp, d weights, d origins / d directions -- they are
// needed long after this launch): what used to be three launches on the serial chain of a training step (pack 5 us, positions 5 us, fill 6 us).
__device__ __forceinline__ void sh16(float dx, float dy, float dz, float* c);
template <bool PACK>
__global__ void __launch_bounds__(256) k_field_prep(FieldK f, float* __restrict__ pack, const float* __restrict__ origins,
                                                    const float* __restrict__ directions, const float* __restrict__ e_bins, int64_t N, int S,
                                                    float4* __restrict__ pos, float* __restrict__ sel, float* __restrict__ shtab,
                                                    float4* __restrict__ zero, int64_t zero_n4, float4* __restrict__ cam_bias, int cam_bias_n4) {
  int bid = blockIdx.x, nblk = gridDim.x;
  if (PACK) {
    const int pkb = PACK_BLOCKS + ((f.num_images < FIELD_MAX_IMAGES ? f.num_images : FIELD_MAX_IMAGES) * 64 + 255) / 256 + PACK_BF_WORDS / 256;  // = pack_blocks()
    if (bid < pkb) { field_pack_body(f, pack, bid * 256 + threadIdx.x); return; }
    bid -= pkb; nblk -= pkb;
  }
  for (int64_t i = bid * (int64_t)blockDim.x + threadIdx.x; i < zero_n4; i += (int64_t)nblk * blockDim.x) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // the backward's per-camera sums start from zero (k_field_emb_finish leaves them zero again; this covers a fresh workspace)
  for (int64_t i = bid * (int64_t)blockDim.x + threadIdx.x; i < cam_bias_n4; i += (int64_t)nblk * blockDim.x) cam_bias[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cam_bias_n4 > 0 && bid == 0 && threadIdx.x == 0) cam_bias[FIELD_MAX_IMAGES * 16] = make_float4(0.f, 0.f, 0.f, 0.f);  // the finishing launch's counter
  const int64_t P = N * (int64_t)S;
  for (int64_t p = bid * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)nblk * blockDim.x) {
    const int64_t ray = tn_div_index(p, S, P);
    const int s = (int)(p - ray * S);
    const float* o = origins + ray * 3;
    const float* d = directions + ray * 3;
    const float* eb = e_bins + ray * (S + 1) + s;
    const Contracted c = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], eb[0], eb[1]);
    const float m = c.sel ? 1.0f : 0.0f;
    pos[p] = make_float4(c.px, c.py, c.pz, m);
    sel[p] = m;
    if (s == 0) {  // the ray's SH16 (a per-ray constant of the colour head's input), in the order the MFMA kernels' lanes hold it
      float sh[16];
      sh16(d[0], d[1], d[2], sh);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int r = 0; r < 8; ++r) shtab[ray * 16 + hh * 8 + r] = sh[(r & 3) + 8 * (r >> 2) + 4 * hh];
    }
  }
}

// ---- encode, XCD-affine and level-major ------------------------------------------------------------------------------------
// The 11 hashed levels of the default grid are 4 MB each: 46 MB that no L2 holds (4 MB per XCD), so a kernel in which every CU gathers from
// every level runs at the chip's random-line rate beyond L2 (55-59 G lines/s: scripts/microbench/gather_rate.hip).  Workgroups are dealt to
// the 8 XCDs round-robin by index, so block b runs on XCD b % 8: here it gathers from ONE level, and an XCD only ever sees the levels of its
// own work list (EncSched: one hashed level at a time, in list order) -- the level's table stays in that XCD's L2 while all samples visit it.
// scripts/microbench/encode_xcd.hip: 104 -> 67 us for 4096 x 48 samples against the lane = (sample, half of the levels) mapping.
// A LANE PAIR per sample: the even lane fetches the four corners with x = ceil, the odd lane those with x = floor.  The two x neighbours of a
// corner pair differ in the low bits of the hashed index only -- same 64-B line 7 times out of 8 -- and as adjacent lanes of ONE load
// instruction they cost the texture path one line instead of two (67 -> 52 us in the microbenchmark).  The pair swaps one component per
// corner over DPP and each lane interpolates ONE of the two features (a + b == b + a exactly, so the odd lane's x lerp  f3 ux + f0 ox  is the
// reference's  f0 ox + f3 ux  bit for bit); outputs are level-major [L][PT] float2 = consecutive floats of the two lanes.
// JAC (training): also res * d enc / d offset (3 floats per lane, planes [L][3 axes][PT] float2 = (feature 0, feature 1)): the backward turns
// d enc into d position with it (k_field_dpos) instead of gathering the 8 x 16 corners again inside the table-gradient scatter.
#ifndef ENC_PER
#define ENC_PER 4        // samples per lane pair
#endif
#define ENC_CHUNK (128 * ENC_PER)  // samples per block
#define ENC_MAX_ITEMS 2  // work items per XCD (TN_MAX_LEVELS / 8)
static_assert(TN_MAX_LEVELS <= 8 * ENC_MAX_ITEMS, "encode plan");
struct EncSched {
  // XCD x runs its items in order; item i covers chunks [c0, c0 + first[i+1] - first[i]) of level `level`
  int16_t level[8][ENC_MAX_ITEMS];
  int32_t first[8][ENC_MAX_ITEMS + 1];  // prefix sum of the items' chunk counts: block q of XCD x belongs to the item with first[i] <= q < first[i+1]
  int32_t c0[8][ENC_MAX_ITEMS];
  int32_t blocks_per_xcd;               // max over x of first[x][last]
};
// Which XCD gathers which (level, chunk range): XCD x runs level x for all samples, then level x + 8.  With the default grid that leaves XCDs
// 5-7 with two hashed levels and the others with one small + one hashed level.  Dealing the chunks out by cost instead (the levels' chunks on
// one line, cut into 8 equal parts: every XCD one whole level plus pieces of others) was measured and LOST: 57.5 -> 58.8-63.0 us -- every
// extra (XCD, level) pair is another 4 MB that an L2 has to fill from cold, and the launch is bound by L2 -> L1 line traffic, not by the
// busiest XCD alone (profiles/r04_experiments.md).
static EncSched make_enc_sched(const TnGrid& g, int64_t P) {
  EncSched sc;
  const int chunks = (int)tn_cdiv(P, ENC_CHUNK);
  int mx = 0;
  for (int x = 0; x < 8; ++x) {
    int n = 0;
    sc.first[x][0] = 0;
    for (int l = x; l < g.num_levels && n < ENC_MAX_ITEMS; l += 8) {
      sc.level[x][n] = (int16_t)l;
      sc.c0[x][n] = 0;
      sc.first[x][n + 1] = sc.first[x][n] + chunks;
      ++n;
    }
    for (int i = n; i < ENC_MAX_ITEMS; ++i) { sc.level[x][i] = -1; sc.c0[x][i] = 0; sc.first[x][i + 1] = sc.first[x][n]; }
    mx = sc.first[x][n] > mx ? sc.first[x][n] : mx;
  }
  sc.blocks_per_xcd = mx;
  return sc;
}
template <bool JAC>
__global__ void __launch_bounds__(256) k_field_encode_xcd(GridK g, EncSched sc, const float4* __restrict__ pos, int64_t P, int64_t PT,
                                                          float* __restrict__ enc, float* __restrict__ jac) {
  const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
  int item = 0;
#pragma unroll
  for (int i = 1; i < ENC_MAX_ITEMS; ++i) item += (q >= sc.first[x][i]) ? 1 : 0;
  if (q >= sc.first[x][ENC_MAX_ITEMS]) return;
  const int l = sc.level[x][item];
  const int chunk = sc.c0[x][item] + (q - sc.first[x][item]);
  const float res = g.res[l];
  const uint32_t off = (uint32_t)l * g.tsize;
  const int side = threadIdx.x & 1;  // 0: the corners with x = ceil and feature 0, 1: x = floor and feature 1
  const int64_t base = (int64_t)chunk * ENC_CHUNK + (threadIdx.x >> 1);
  float2 fv[ENC_PER][4];
  float ox[ENC_PER], oy[ENC_PER], oz[ENC_PER];
#pragma unroll
  for (int u = 0; u < ENC_PER; ++u) {
    int64_t p = base + u * 128;
    if (p >= P) p = P - 1;
    const float4 c = pos[p];
    // tn_level_corners, this lane's half: slots 0..3 = (x, c, c) (x, f, c) (x, c, f) (x, f, f)  <->  f0|f3, f1|f2, f4|f7, f5|f6
    const float sx = c.x * res, sy = c.y * res, sz = c.z * res;
    const float fxf = floorf(sx), fyf = floorf(sy), fzf = floorf(sz);
    const uint32_t xi = side ? (uint32_t)(int)fxf : (uint32_t)(int)ceilf(sx);
    const uint32_t hcy = (uint32_t)(int)ceilf(sy) * TN_PRIME_Y, hfy = (uint32_t)(int)fyf * TN_PRIME_Y;
    const uint32_t hcz = (uint32_t)(int)ceilf(sz) * TN_PRIME_Z, hfz = (uint32_t)(int)fzf * TN_PRIME_Z;
    ox[u] = sx - fxf; oy[u] = sy - fyf; oz[u] = sz - fzf;
    fv[u][0] = tn_table_entry(g.table, ((xi ^ hcy ^ hcz) & g.mask) + off);
    fv[u][1] = tn_table_entry(g.table, ((xi ^ hfy ^ hcz) & g.mask) + off);
    fv[u][2] = tn_table_entry(g.table, ((xi ^ hcy ^ hfz) & g.mask) + off);
    fv[u][3] = tn_table_entry(g.table, ((xi ^ hfy ^ hfz) & g.mask) + off);
  }
#pragma unroll
  for (int u = 0; u < ENC_PER; ++u) {
    const int64_t p = base + u * 128;
    const float ux = 1.0f - ox[u], uy = 1.0f - oy[u], uz = 1.0f - oz[u];
    const float wk = side ? ux : ox[u], wr = side ? ox[u] : ux;  // weights of the kept corner (own x) and the received one (the partner's x)
    float t[4], dk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float keep = side ? fv[u][k].y : fv[u][k].x, send = side ? fv[u][k].x : fv[u][k].y;
      const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
      t[k] = keep * wk + recv * wr;                     // f03, f12, f47, f56 of this lane's feature
      if (JAC) dk[k] = side ? recv - keep : keep - recv;  // f0 - f3, f1 - f2, f4 - f7, f5 - f6
    }
    const float f0312 = t[0] * oy[u] + t[1] * uy, f4756 = t[2] * oy[u] + t[3] * uy;
    const float r = f0312 * oz[u] + f4756 * uz;
    if (p >= P) continue;
    enc[((int64_t)l * PT + p) * 2 + side] = r;
    if (JAC) {
      const float dx = (dk[0] * oy[u] + dk[1] * uy) * oz[u] + (dk[2] * oy[u] + dk[3] * uy) * uz;
      const float dy = (t[0] - t[1]) * oz[u] + (t[2] - t[3]) * uz;
      const float dz = f0312 - f4756;
      float* jp = jac + ((int64_t)l * 3 * PT + p) * 2 + side;
      __builtin_nontemporal_store(dx * res, jp);
      __builtin_nontemporal_store(dy * res, jp + 2 * PT);
      __builtin_nontemporal_store(dz * res, jp + 4 * PT);
    }
  }
}

// (emb_finish_jobs, field_dpos_body: tn_field_dpos.h)
__global__ void __launch_bounds__(256) k_field_emb_finish(float* __restrict__ cam_bias, uint32_t* __restrict__ counter, const float* __restrict__ hw0,
                                                          const float* __restrict__ emb, float* __restrict__ gemb, float* __restrict__ ghw0, int num_images) {
  __shared__ float sums[64];
  __shared__ int flag;
  emb_finish_jobs(cam_bias, counter, hw0, emb, gemb, ghw0, num_images, sums, &flag, blockIdx.x, gridDim.x);
}

// d position of every sample (field_dpos_body) as a launch of its own
__global__ void __launch_bounds__(256) k_field_dpos(DposArgs a) {
  __shared__ float emb_sums[64];
  __shared__ int emb_flag;
  field_dpos_body(a, blockIdx.x, gridDim.x, emb_sums, &emb_flag);
}

// ---- SH degree 4 on (d+1)/2, un-remapped (utils/math.py:45-78; fields/base_field.py:136-142) -------------------------------
__device__ __forceinline__ void sh16(float dx, float dy, float dz, float* c) {
  float x = (dx + 1.0f) / 2.0f, y = (dy + 1.0f) / 2.0f, z = (dz + 1.0f) / 2.0f;
  float xx = x * x, yy = y * y, zz = z * z;
  c[0] = 0.28209479177387814f;
  c[1] = 0.4886025119029199f * y;
  c[2] = 0.4886025119029199f * z;
  c[3] = 0.4886025119029199f * x;
  c[4] = 1.0925484305920792f * x * y;
  c[5] = 1.0925484305920792f * y * z;
  c[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
  c[7] = 1.0925484305920792f * x * z;
  c[8] = 0.5462742152960396f * (xx - yy);
  c[9] = 0.5900435899266435f * y * (3.0f * xx - yy);
  c[10] = 2.890611442640554f * x * y * z;
  c[11] = 0.4570457994644658f * y * (5.0f * zz - 1.0f);
  c[12] = 0.3731763325901154f * z * (5.0f * zz - 3.0f);
  c[13] = 0.4570457994644658f * x * (5.0f * zz - 1.0f);
  c[14] = 1.445305721320277f * z * (xx - yy);
  c[15] = 0.5900435899266435f * x * (xx - 3.0f * yy);
}

// ---- MFMA helpers ------------------------------------------------------------------------------------------------------
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// ---- split-bf16 products (TN_HEAD_BF16X3=1, the colour head only; see PACK_BF_OFF) ----------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMAB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// registers 8s..8s+7 of a D-layout tile as the B (or A) operand of k-step s: x = hi + lo (+ ~2^-17 |x|)
template <int S8>
__device__ __forceinline__ void bf_split8(const f32x16& v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float x = v[8 * S8 + q];
    const __bf16 hq = (__bf16)x;
    hi[q] = hq;
    lo[q] = (__bf16)(x - (float)hq);
  }
}
// o0 / o1 += W[0..31 / 32..63][32 T .. 32 T + 31] . x  for one 32-feature input tile x (D-layout), fragments of layer LAYER at `frags`
// ([m][t][s][part][lane] of 4 words); three products per k-step, the small ones first
template <int T>
__device__ __forceinline__ void bf3_tile_fwd(const float* __restrict__ frags, int lane, const f32x16& x, f32x16& o0, f32x16& o1) {
  bf16x8 bh, bl;
#define BF_FRAG(m, s, part) (*reinterpret_cast<const bf16x8*>(frags + ((((((m) * 2 + T) * 2 + (s)) * 2 + (part))) << 8) + (lane << 2)))
#define BF_KSTEP(s)                                                                       \
  {                                                                                       \
    const bf16x8 a0h = BF_FRAG(0, s, 0), a0l = BF_FRAG(0, s, 1), a1h = BF_FRAG(1, s, 0), a1l = BF_FRAG(1, s, 1); \
    o0 = MFMAB(a0l, bh, o0); o1 = MFMAB(a1l, bh, o1);                                     \
    o0 = MFMAB(a0h, bl, o0); o1 = MFMAB(a1h, bl, o1);                                     \
    o0 = MFMAB(a0h, bh, o0); o1 = MFMAB(a1h, bh, o1);                                     \
  }
  bf_split8<0>(x, bh, bl);
  BF_KSTEP(0)
  bf_split8<1>(x, bh, bl);
  BF_KSTEP(1)
#undef BF_KSTEP
#undef BF_FRAG
}

__device__ __forceinline__ f32x16 bias_tile(const float* __restrict__ lds_bias, int layer, int m, int h) {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = lds_bias[layer * 64 + 32 * m + RROW(r, h)];
  return v;
}
__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
  return v;
}
// row-major [P][ld] <-> D-layout tile m: registers 4g..4g+3 <-> features 32m + 8g + 4h + {0..3}
__device__ __forceinline__ void store_tile(float* __restrict__ base, int64_t p, int ld, int m, int h, const f32x16& v) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(base + p * ld + 32 * m + 8 * g + 4 * h) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
__device__ __forceinline__ f32x16 load_tile(const float* __restrict__ base, int64_t p, int ld, int m, int h) {
  f32x16 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 t = *reinterpret_cast<const float4*>(base + p * ld + 32 * m + 8 * g + 4 * h);
    v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
  }
  return v;
}

// fragment order (see FieldWs): tile m of an activation with M 32-feature tiles per sample
// Non-temporal on both sides: the activations are written once and read once (600 MB per step at 4096 rays), the hash tables are gathered from
// again and again -- streaming the activations past the caches keeps the tables in the Infinity Cache (the Adam kernels do the same with moments).
__device__ __forceinline__ void store_frag(float* __restrict__ base, int64_t tile, int M, int m, int lane, const f32x16& v) {
  v4f_t* b = reinterpret_cast<v4f_t*>(base) + ((tile * M + m) * 4) * 64 + lane;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const v4f_t t = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    if (FRAG_NT) __builtin_nontemporal_store(t, b + g * 64); else b[g * 64] = t;
  }
}
__device__ __forceinline__ f32x16 load_frag(const float* __restrict__ base, int64_t tile, int M, int m, int lane) {
  const v4f_t* b = reinterpret_cast<const v4f_t*>(base) + ((tile * M + m) * 4) * 64 + lane;
  f32x16 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const v4f_t t = FRAG_NT ? __builtin_nontemporal_load(b + g * 64) : b[g * 64];
    v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
  }
  return v;
}
// the 16 base-MLP outputs of a tile (registers 0..7 of the Linear(64,16) accumulator: rows R(r,h) < 16) as [tile][2][lane] float4
__device__ __forceinline__ void store_bo(float* __restrict__ base, int64_t tile, int lane, const f32x16& bo) {
  v4f_t* b = reinterpret_cast<v4f_t*>(base) + tile * 128 + lane;
  const v4f_t t0 = {bo[0], bo[1], bo[2], bo[3]}, t1 = {bo[4], bo[5], bo[6], bo[7]};
  if (FRAG_NT) { __builtin_nontemporal_store(t0, b); __builtin_nontemporal_store(t1, b + 64); }
  else { b[0] = t0; b[64] = t1; }
}
// the density logit = head-input slot 16 = base output row 0 = group 0, half 0, component 0 of sample j
__device__ __forceinline__ int64_t hin_logit_index(int64_t tile, int j) { return (tile * 128 + j) * 4; }
// level-major encoding -> the D-layout of the first layer's B operand: registers 4q .. 4q+3 = levels 4q + 2h (f0, f1), 4q + 2h + 1 (f0, f1)
// (one 32-bit lane offset for all eight loads on top of wave-uniform bases: a 64-bit address per load costs the one-wave-per-SIMD backward
// kernel 16 registers it does not have.  (2 PT + P) * 8 < 2^32 is checked on the host.)
__device__ __forceinline__ f32x16 load_enc_lm(const float* __restrict__ enc, int64_t PT, int64_t pc, int h, int L) {
  f32x16 v;
  const uint32_t voff = (uint32_t)(((int64_t)(2 * h) * PT + pc) * 8);
  const char* e = reinterpret_cast<const char*>(enc);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int l0 = 4 * q + 2 * h;
    const char* b0 = e + (int64_t)(4 * q) * PT * 8;  // wave-uniform
    const float2 a = l0 < L ? *reinterpret_cast<const float2*>(b0 + voff) : make_float2(0.f, 0.f);
    const float2 b = l0 + 1 < L ? *reinterpret_cast<const float2*>(b0 + PT * 8 + voff) : make_float2(0.f, 0.f);
    v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = b.x; v[4 * q + 3] = b.y;
  }
  return v;
}

// ---- forward MLP chain ----------------------------------------------------------------------------------------------
// The gather is k_field_encode_xcd's; this kernel is the chain alone.  Round 3 ran both in one launch (lane = (sample, half of the levels), the
// encoding went from the gather straight into the first MFMA): 145 us at 4096 x 48 samples, of which the gather alone -- at the chip's
// random-line rate beyond L2 -- is 104 us and the 224 MFMAs per tile 36 us: with two waves per SIMD the two hardly overlapped.  With the
// XCD-affine encode (67 us) the chain reads 25 MB of encoding back and is bound by its MFMAs and, in training, its activation stores.
// TRAIN keeps what the backward needs: relu(h1), the 16 base outputs, relu(hh1), relu(hh2), the sigmoid outputs.  SH16 and the appearance
// embedding (48 of the 64 head-input slots) are per-ray constants: the backward rebuilds them (bit-identical: same code).
#ifndef FWD_THREADS
#define FWD_THREADS 256
#endif
// FWD_ABLATE (compile-time, timing diagnostics only -- results are wrong with any bit set): 1 no activation stores, 2 no per-ray loads (sh table,
// camhead row), 4 no encoding loads, 8 no last layer / sigmoid / output stores, 16 no density / base-output stores (scripts/fwd_ablation.sh)
#ifndef FWD_ABLATE
#define FWD_ABLATE 0
#endif
// FWD_HEAD2_MFMA=1: the colour head's last layer as a 32x32x2 MFMA tile (round 3's form; A/B aid)
#ifndef FWD_HEAD2_MFMA
#define FWD_HEAD2_MFMA 0
#endif
// FWD_EMB_MFMA=1: the camera's embedding through the matrix cores in training too (A/B aid)
#ifndef FWD_EMB_MFMA
#define FWD_EMB_MFMA 0
#endif
template <bool TRAIN, bool BF3>
__global__ void __launch_bounds__(FWD_THREADS, FWD_THREADS / 128) k_field_mlp_fwd(const float* __restrict__ pack, const float* __restrict__ encs, int L,
                                                            int64_t PT, const float* __restrict__ sels, const float* __restrict__ shtab,
                                                            const int64_t* __restrict__ cam_idx, const float* __restrict__ emb, int num_images,
                                                            int use_cam_emb, int64_t N, int S, int C, float* __restrict__ density,
                                                            float* __restrict__ rgb, float* __restrict__ density_pre, float* __restrict__ h1s,
                                                            float* __restrict__ hins, float* __restrict__ hh1s, float* __restrict__ hh2s,
                                                            float* __restrict__ ys) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // PACK_FWD_TOTAL floats
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int64_t P = N * (int64_t)S;
  const int64_t ntiles = tn_cdiv(P, TILE);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  f32x16 nin;  // the next tile's encoding, requested one tile ahead -- the first one before the weights are staged (a wave makes three trips)
  if (wave < ntiles) {
    const int64_t p0 = wave * TILE + j;
    nin = (FWD_ABLATE & 4) ? f32x16{0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f, 0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f} : load_enc_lm(encs, PT, p0 < P ? p0 : P - 1, h, L);
  }
  for (int i = threadIdx.x * 4; i < PACK_FWD_TOTAL; i += blockDim.x * 4) {
    // (split-bf16 head: the two 64-wide head layers' fragments -- the same 8192 words -- come from the bf16 region instead)
    const float* src = (BF3 && i >= fwd_off(2) && i < fwd_off(4)) ? pack + PACK_BF_OFF + (i - fwd_off(2)) : pack + i;
    *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(src);
  }
  __syncthreads();
  const float* lbias = lds + PACK_BIAS_OFF;
#define AF(layer, m, t, r) lds[fwd_off(layer) + ((((m) * layer_ti(layer) + (t)) * 16 + (r)) << 6) + lane]
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t p = tile * TILE + j;
    const bool valid = p < P;
    const int64_t pc = valid ? p : P - 1;
    const int64_t ray = tn_div_index(pc, S, P);
    const float sl = sels[pc];
    const f32x16 in0 = nin;
    // ---------------- base MLP: Linear(32,64) ReLU Linear(64,16)
    f32x16 a0 = bias_tile(lbias, 0, 0, h), a1 = bias_tile(lbias, 0, 1, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a0 = MFMA(AF(0, 0, 0, r), in0[r], a0);
      a1 = MFMA(AF(0, 1, 0, r), in0[r], a1);
    }
    {
      const int64_t tn = tile + nwaves < ntiles ? tile + nwaves : tile;
      const int64_t pn = tn * TILE + j;
      if (!(FWD_ABLATE & 4)) nin = load_enc_lm(encs, PT, pn < P ? pn : P - 1, h, L);
    }
    __builtin_amdgcn_sched_barrier(0);
    a0 = relu16(a0); a1 = relu16(a1);
    if (TRAIN && !(FWD_ABLATE & 1)) { store_frag(h1s, tile, 2, 0, lane, a0); store_frag(h1s, tile, 2, 1, lane, a1); }
    f32x16 bo = bias_tile(lbias, 1, 0, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) bo = MFMA(AF(1, 0, 0, r), a0[r], bo);
#pragma unroll
    for (int r = 0; r < 16; ++r) bo = MFMA(AF(1, 0, 1, r), a1[r], bo);
    __builtin_amdgcn_sched_barrier(0);
    if (valid && h == 0 && !(FWD_ABLATE & 16)) {
      float pre = bo[0];
      density[p] = expf(pre) * sl;  // average_init_density (=1.0) * trunc_exp(pre) * selector
      if (density_pre) density_pre[p] = pre;
    }
    if (TRAIN && !(FWD_ABLATE & 1)) store_bo(hins, tile, lane, bo);  // whole tiles: lanes past the end hold the (finite) values of the last sample
    // ---------------- head input in slot space: tile 0 = sh[R(r,h)] (r<8) | base_out rows (r>=8), tile 1 = embedding
    f32x16 hi0, hi1;
    {
      const float4 zz = make_float4(0.1f, 0.2f, 0.3f, 0.4f);
      const float4 s0 = (FWD_ABLATE & 2) ? zz : *reinterpret_cast<const float4*>(shtab + ray * 16 + h * 8), s1 = (FWD_ABLATE & 2) ? zz : *reinterpret_cast<const float4*>(shtab + ray * 16 + h * 8 + 4);
      hi0[0] = s0.x; hi0[1] = s0.y; hi0[2] = s0.z; hi0[3] = s0.w; hi0[4] = s1.x; hi0[5] = s1.y; hi0[6] = s1.z; hi0[7] = s1.w;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) hi0[8 + r] = bo[r];
    f32x16 c0 = bias_tile(lbias, 2, 0, h), c1 = bias_tile(lbias, 2, 1, h);
    constexpr bool CAMHEAD = TRAIN && !FWD_EMB_MFMA;  // (the training launch is the one with per-camera embeddings: use_cam_emb == TRAIN)
    if (CAMHEAD && !(FWD_ABLATE & 2)) {
      // the camera's embedding enters head layer 0 as the per-camera vector camhead[cam] (built with the fragments): no MFMA for slots 32..63
      int64_t cam = cam_idx[ray];
      if (cam < 0 || cam >= num_images) cam = 0;
      const float* chp = pack + PACK_CAMHEAD_OFF + cam * 64 + 4 * h;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 u0 = *reinterpret_cast<const float4*>(chp + 8 * gq), u1 = *reinterpret_cast<const float4*>(chp + 32 + 8 * gq);
        c0[4 * gq] += u0.x; c0[4 * gq + 1] += u0.y; c0[4 * gq + 2] += u0.z; c0[4 * gq + 3] += u0.w;
        c1[4 * gq] += u1.x; c1[4 * gq + 1] += u1.y; c1[4 * gq + 2] += u1.z; c1[4 * gq + 3] += u1.w;
      }
    } else if (!CAMHEAD) {
      const float* ebp;
      if (use_cam_emb) {
        int64_t cam = cam_idx[ray];
        if (cam < 0 || cam >= num_images) cam = 0;
        ebp = emb + cam * 32;
      } else {
        ebp = lds + PACK_MEANEMB_OFF;  // ones * embedding.mean(0)  (fields/nerfacto_field.py:292-295)
      }
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int q = 0; q < 4; ++q) hi1[4 * gq + q] = ebp[8 * gq + 4 * h + q];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- head layer 0
    if (BF3) {
      bf3_tile_fwd<0>(lds + fwd_off(2), lane, hi0, c0, c1);
      if (!CAMHEAD && !(FWD_ABLATE & 2)) bf3_tile_fwd<1>(lds + fwd_off(2), lane, hi1, c0, c1);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { c0 = MFMA(AF(2, 0, 0, r), hi0[r], c0); c1 = MFMA(AF(2, 1, 0, r), hi0[r], c1); }
      if (!CAMHEAD && !(FWD_ABLATE & 2)) {  // (inference: the mean embedding goes through the matrix cores as before)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0 = MFMA(AF(2, 0, 1, r), hi1[r], c0); c1 = MFMA(AF(2, 1, 1, r), hi1[r], c1); }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    c0 = relu16(c0); c1 = relu16(c1);
    if (TRAIN && !(FWD_ABLATE & 1)) { store_frag(hh1s, tile, 2, 0, lane, c0); store_frag(hh1s, tile, 2, 1, lane, c1); }
    // ---------------- head layer 1
    f32x16 d0 = bias_tile(lbias, 3, 0, h), d1 = bias_tile(lbias, 3, 1, h);
    if (BF3) {
      bf3_tile_fwd<0>(lds + fwd_off(2) + PACK_BF_LAYER_WORDS, lane, c0, d0, d1);
      bf3_tile_fwd<1>(lds + fwd_off(2) + PACK_BF_LAYER_WORDS, lane, c1, d0, d1);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { d0 = MFMA(AF(3, 0, 0, r), c0[r], d0); d1 = MFMA(AF(3, 1, 0, r), c0[r], d1); }
#pragma unroll
      for (int r = 0; r < 16; ++r) { d0 = MFMA(AF(3, 0, 1, r), c1[r], d0); d1 = MFMA(AF(3, 1, 1, r), c1[r], d1); }
    }
    __builtin_amdgcn_sched_barrier(0);
    d0 = relu16(d0); d1 = relu16(d1);
    if (TRAIN && !(FWD_ABLATE & 1)) { store_frag(hh2s, tile, 2, 0, lane, d0); store_frag(hh2s, tile, 2, 1, lane, d1); }
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- head layer 2 + sigmoid.  Linear(64, C <= 4) on the VECTOR ALU: as a 32-row MFMA tile it spent 32 matrix instructions
    // (2 064 cycles of the 14.4 k per tile) on 4 useful rows.  Lane (j, h) holds 32 of its sample's 64 inputs (tiles 0 / 1, registers r <->
    // features 32 t + R(r, h)); the weights of the four outputs for that feature are lanes 32 h + 0..3 of the layer's forward fragment
    // (Af[4][0][t][r][32 h + c] = W[c][32 t + R(r, h)]): one broadcast ds_read_b128 per (t, r), four FMAs, then the two halves are added.
    float e[4] = {0.f, 0.f, 0.f, 0.f};
    if (FWD_ABLATE & 8) { if (d0[0] + d1[5] == 1.2345e-30f) rgb[p] = d0[3]; continue; }
#if FWD_HEAD2_MFMA
    {
      f32x16 em = bias_tile(lbias, 4, 0, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) em = MFMA(AF(4, 0, 0, r), d0[r], em);
#pragma unroll
      for (int r = 0; r < 16; ++r) em = MFMA(AF(4, 0, 1, r), d1[r], em);
      e[0] = em[0]; e[1] = em[1]; e[2] = em[2]; e[3] = em[3];
    }
#else
    {
      const float* w4 = lds + fwd_off(4) + 32 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 wa = *reinterpret_cast<const float4*>(w4 + (r << 6)), wb = *reinterpret_cast<const float4*>(w4 + ((16 + r) << 6));
        e[0] = fmaf(wa.x, d0[r], e[0]); e[1] = fmaf(wa.y, d0[r], e[1]); e[2] = fmaf(wa.z, d0[r], e[2]); e[3] = fmaf(wa.w, d0[r], e[3]);
        e[0] = fmaf(wb.x, d1[r], e[0]); e[1] = fmaf(wb.y, d1[r], e[1]); e[2] = fmaf(wb.z, d1[r], e[2]); e[3] = fmaf(wb.w, d1[r], e[3]);
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) e[cc] = (e[cc] + __shfl_xor(e[cc], 32, 64)) + lbias[4 * 64 + cc];
    }
#endif
    if (valid && h == 0) {
      float y[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        if (cc < C) { y[cc] = 1.0f / (1.0f + expf(-e[cc])); rgb[p * C + cc] = y[cc]; }
      if (TRAIN) *reinterpret_cast<float4*>(ys + p * 4) = make_float4(y[0], y[1], y[2], y[3]);
    }
  }
#undef AF
}

// density only (no head): cross-evaluated density2 / density2_thermal.  TRAIN keeps what the density-only backward needs: relu(layer 0) and
// the density logit (in head-input slot 16, where the full forward leaves it).
template <bool TRAIN>
__global__ void __launch_bounds__(256, 2) k_field_density_only(const float* __restrict__ pack, const float* __restrict__ enc, int L, int64_t PT,
                                                               const float* __restrict__ sel, int64_t P, float* __restrict__ density,
                                                               float* __restrict__ h1s, float* __restrict__ hins) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x * 4; i < PACK_FWD_TOTAL; i += blockDim.x * 4)
    *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(pack + i);
  __syncthreads();
  const float* lbias = lds + PACK_BIAS_OFF;
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int64_t ntiles = tn_cdiv(P, TILE);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
#define AF(layer, m, t, r) lds[fwd_off(layer) + ((((m) * layer_ti(layer) + (t)) * 16 + (r)) << 6) + lane]
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    int64_t p = tile * TILE + j;
    bool valid = p < P;
    int64_t pc = valid ? p : P - 1;
    f32x16 in0 = load_enc_lm(enc, PT, pc, h, L);
    f32x16 a0 = bias_tile(lbias, 0, 0, h), a1 = bias_tile(lbias, 0, 1, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0 = MFMA(AF(0, 0, 0, r), in0[r], a0); a1 = MFMA(AF(0, 1, 0, r), in0[r], a1); }
    a0 = relu16(a0); a1 = relu16(a1);
    if (TRAIN) { store_frag(h1s, tile, 2, 0, lane, a0); store_frag(h1s, tile, 2, 1, lane, a1); }
    f32x16 bo = bias_tile(lbias, 1, 0, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) bo = MFMA(AF(1, 0, 0, r), a0[r], bo);
#pragma unroll
    for (int r = 0; r < 16; ++r) bo = MFMA(AF(1, 0, 1, r), a1[r], bo);
    if (valid && h == 0) {
      density[p] = expf(bo[0]) * sel[p];
      if (TRAIN) hins[hin_logit_index(tile, j)] = bo[0];
    }
  }
#undef AF
}

// ---- backward MLP chain WITH the weight gradients (the default) ------------------------------------------------------------------
// k_field_bwd_fused = k_field_mlp_bwd + k_wgrad_batch in one launch: the pre-activation gradients never leave the chip.
//   Before: k_field_mlp_bwd wrote 242 MB of pre-activation gradients, k_wgrad_batch (companion stream) read them back with the saved
//   activations (~450 MB) while k_grid_bin / k_grid_fold ran beside it -- 460 us in-step for the weight gradients and the table scatter
//   stretched from 330 to 529 us (LDS pipe + HBM shared).
//   Now: one wave = one 32-sample tile at a time, ONE wave per SIMD (__launch_bounds__(256, 1): the weight-gradient accumulators of all five
//   layers live in registers for the whole kernel -- 64 + 64 + 32 + 16 + 16 = 192 of the 512).  Per layer the wave
//     1. runs the chain step dIN^T = W^T . dOUT^T (accumulator tile = next B operand, as before),
//     2. writes dY (just computed) and X (the saved activation, loaded for the ReLU mask anyway) of the tile to its private LDS buffers,
//        sample-major with a 68-float row stride (conflict-free ds_write_b128 AND ds_read_b32),
//     3. reads them back TRANSPOSED (lane = feature, k = sample) as the A / B operands of dW += dY^T X:
//        v_mfma_f32_32x32x2_f32 for the 64-wide layers, v_mfma_f32_16x16x4_f32 for the two layers with <= 16 outputs (Linear(64,16), Linear(64,C)).
//   Bias gradients are the lane sums of the A operands; the appearance-embedding rows are hw0[:, 31:63]^T times head layer 0's bias gradient
//   restricted to the camera: running per-camera sums (lane = feature; up to FIELD_MAX_IMAGES cameras, two 128-B atomic segments per camera
//   change) into cam_bias, multiplied by hw0 in k_field_emb_finish (one 64-thread block per camera behind this launch).
//   Epilogue: the block's four waves add their accumulators in LDS (plain read-modify-write, one wave per turn) and ONE burst of float
//   atomics per block goes to the gradient arena (256 blocks x 12.5 k floats).
//   MFMA work per tile: chain 152 + weight gradients 128 (32x32x2) + 64 (16x16x4, half the cycles) -> 20.2 k cycles; 6 tiles per wave at 4096 rays.
#define TSTR 68                       // LDS row stride (floats) of a [32 samples][<= 64 features] transposition tile
#define FB_TILE_FLOATS (32 * TSTR)    // one tile buffer
#define FB_WAVE_FLOATS (2 * FB_TILE_FLOATS + 128 + 32)  // dY tile, X tile, g3 [32][4], camera of each sample [32]
#define FB_LDS_FLOATS (PACK_BWD_FLOATS + 4 * FB_WAVE_FLOATS)
// block-sum layout (floats, inside the per-wave region): W3 [64][64] | W2 [64 out][32 slots] (+ 2048 unused) | W0 [64][32] | W4 [16][64] | W1 [16][64] | b3 b2 b0 (64 each) | b4 b1 (16 each)
#define FB_RED_W3 0
#define FB_RED_W2 4096
#define FB_RED_W0 8192
#define FB_RED_W4 10240
#define FB_RED_W1 11264
#define FB_RED_B3 12288
#define FB_RED_B2 12352
#define FB_RED_B0 12416
#define FB_RED_B4 12480
#define FB_RED_B1 12496
#define FB_RED_TOTAL 12512
static_assert(FB_RED_TOTAL + 4 * 64 + 8 <= 4 * FB_WAVE_FLOATS, "block-sum area (+ the embedding merge) must fit in the per-wave buffers");
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define WAVE_LDS_SYNC()                                   \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

// D-layout tile m -> LDS tile (sample-major): registers 4g..4g+3 <-> features 32m + 8g + 4h + {0..3} of sample j
__device__ __forceinline__ void lds_put_tile(float* __restrict__ buf, int j, int h, int m, const f32x16& v) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(buf + j * TSTR + 32 * m + 8 * g + 4 * h) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
// dW[MO x 32][MI x 32] += dY^T X over the tile's 32 samples; bsum[a] += this lane's share of sum_p dY[p][32a + j]
template <int MO, int MI>
__device__ __forceinline__ void wgrad_tile32(const float* __restrict__ bufY, const float* __restrict__ bufX, int j, int h, f32x16 (&acc)[MO][MI],
                                             float (&bsum)[MO]) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    float av[MO], bv[MI];
#pragma unroll
    for (int a = 0; a < MO; ++a) av[a] = bufY[(2 * t + h) * TSTR + 32 * a + j];
#pragma unroll
    for (int b = 0; b < MI; ++b) bv[b] = bufX[(2 * t + h) * TSTR + 32 * b + j];
#pragma unroll
    for (int a = 0; a < MO; ++a) {
      bsum[a] += av[a];
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = MFMA(av[a], bv[b], acc[a][b]);
    }
  }
}
// dW[16][64] += dY^T X with dY [32 samples][ldy >= 16 valid columns, zero beyond ncols] ; lane l: row/col l % 16, k-slice l / 16
__device__ __forceinline__ void wgrad_tile16(const float* __restrict__ bufY, int ldy, int ncols, const float* __restrict__ bufX, int lane,
                                             f32x4v (&acc)[4], float& bsum) {
  const int c = lane & 15, ks = lane >> 4;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float av = c < ncols ? bufY[(4 * t + ks) * ldy + c] : 0.0f;
    bsum += av;
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = MFMA16(av, bufX[(4 * t + ks) * TSTR + 16 * b + c], acc[b]);
  }
}

// ---- split-bf16 forms of the colour head's backward products (TN_HEAD_BF16X3=1) ------------------------------------------------------------
// o += W[rows of output tile 0][32 M .. 32 M + 31] . x for one 32-feature tile x (fragments [t][m][s][part], t = 0)
template <int M>
__device__ __forceinline__ void bf3_tile_one(const float* __restrict__ frags, int lane, const f32x16& x, f32x16& o) {
  bf16x8 bh, bl;
#define BF_FRAG(s, part) (*reinterpret_cast<const bf16x8*>(frags + ((((M * 2 + (s)) * 2 + (part))) << 8) + (lane << 2)))
  bf_split8<0>(x, bh, bl);
  { const bf16x8 ah = BF_FRAG(0, 0), al = BF_FRAG(0, 1); o = MFMAB(al, bh, o); o = MFMAB(ah, bl, o); o = MFMAB(ah, bh, o); }
  bf_split8<1>(x, bh, bl);
  { const bf16x8 ah = BF_FRAG(1, 0), al = BF_FRAG(1, 1); o = MFMAB(al, bh, o); o = MFMAB(ah, bl, o); o = MFMAB(ah, bh, o); }
#undef BF_FRAG
}
// wgrad_tile32 on split bf16: the tile's 32 samples are TWO k-steps of 16; lane (j, h) reads samples 16 s + 8 h + q (q = 0..7) of its column
// from the same sample-major tiles, splits them, and every (output tile, input tile) pair takes three products per k-step
template <int MO, int MI>
__device__ __forceinline__ void wgrad_tile32_bf3(const float* __restrict__ bufY, const float* __restrict__ bufX, int j, int h, f32x16 (&acc)[MO][MI],
                                                 float (&bsum)[MO]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 ah[MO], al[MO], bh[MI], bl[MI];
#pragma unroll
    for (int a = 0; a < MO; ++a) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float x = bufY[(16 * s + 8 * h + q) * TSTR + 32 * a + j];
        bsum[a] += x;
        const __bf16 hq = (__bf16)x;
        ah[a][q] = hq;
        al[a][q] = (__bf16)(x - (float)hq);
      }
    }
#pragma unroll
    for (int b = 0; b < MI; ++b) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float x = bufX[(16 * s + 8 * h + q) * TSTR + 32 * b + j];
        const __bf16 hq = (__bf16)x;
        bh[b][q] = hq;
        bl[b][q] = (__bf16)(x - (float)hq);
      }
    }
#pragma unroll
    for (int a = 0; a < MO; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) {
        acc[a][b] = MFMAB(al[a], bh[b], acc[a][b]);
        acc[a][b] = MFMAB(ah[a], bl[b], acc[a][b]);
        acc[a][b] = MFMAB(ah[a], bh[b], acc[a][b]);
      }
  }
}

struct FusedGrads {
  float *gw0, *gb0, *gw1, *gb1, *ghw0, *ghb0, *ghw1, *ghb1, *ghw2, *ghb2, *gemb;
};

// FB_ABLATE (compile-time, timing diagnostics only -- results are wrong with any bit set): 1 embedding rows, 2 32x32 weight gradients,
// 4 16x16 weight gradients, 8 block sum + atomics
#ifndef FB_ABLATE
#define FB_ABLATE 0
#endif
// what a tile needs besides the big activation tiles (loaded one tile ahead)
struct FbSmall {
  float pre_logit, g_dens, sel_p;  // density logit (head-input slot 16), d density, selector
  float g3[4];                     // sigmoid backward (half 0 only)
  int cam;                         // camera of this lane's sample, -1 beyond the last sample
  int ray;                         // ray of this lane's sample (the last ray beyond the last sample)
};
template <bool DENS_ONLY>
__device__ __forceinline__ FbSmall fb_load_small(int64_t tile, int64_t tile_addr, int j, int h, int64_t P, int S, int C, int num_images,
                                                 const float* __restrict__ hins, const float* __restrict__ d_density, const float* __restrict__ sel,
                                                 const float* __restrict__ ys, const float* __restrict__ d_rgb, const int64_t* __restrict__ cam_idx) {
  // tile: the tile whose samples these are (beyond the last tile: no valid sample, every gradient zero); tile_addr <= the last tile: where the
  // per-tile array is read (k_field_bwd_fused passes the same index twice)
  FbSmall q;
  const int64_t p = tile * TILE + j;
  const bool valid = p < P;
  const int64_t pc = valid ? p : P - 1;
  q.pre_logit = hins[hin_logit_index(tile_addr, j)];
  q.g_dens = valid ? d_density[pc] : 0.0f;
  q.sel_p = sel[pc];
  q.g3[0] = q.g3[1] = q.g3[2] = q.g3[3] = 0.0f;
  q.cam = -1;
  q.ray = 0;
  if (!DENS_ONLY) {
    q.ray = (int)tn_div_index(pc, S, P);
    const int64_t cam = cam_idx[q.ray];
    q.cam = valid ? ((cam < 0 || cam >= num_images) ? 0 : (int)cam) : -1;
    if (valid && h == 0) {
      const float4 y4 = *reinterpret_cast<const float4*>(ys + p * 4);
      const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C) q.g3[c] = d_rgb[p * C + c] * yv[c] * (1.0f - yv[c]);
    }
  }
  return q;
}

#ifndef FB_NO_SB
#define FB_NO_SB 0
#endif
#ifndef FB_TOUCH
#define FB_TOUCH 0  // L2 warm-up loads of the next tile (compile-time experiment, see the kernel): 127 -> 132 us, off
#endif
#define FB_SB() do { if (!FB_NO_SB) __builtin_amdgcn_sched_barrier(0); } while (0)
template <bool DENS_ONLY, bool BF3>
__global__ void __launch_bounds__(256, 1) k_field_bwd_fused(const float* __restrict__ pack, const float* __restrict__ sel, const float* __restrict__ ys,
                                                            const float* __restrict__ d_rgb, const float* __restrict__ d_density,
                                                            const int64_t* __restrict__ cam_idx, int num_images, int64_t P, int S, int C,
                                                            const float* __restrict__ shtab, const float* __restrict__ emb,
                                                            float* __restrict__ cam_bias, int L, int64_t PT,
                                                            const float* __restrict__ encs, const float* __restrict__ h1s,
                                                            const float* __restrict__ hins, const float* __restrict__ hh1s,
                                                            const float* __restrict__ hh2s, float* __restrict__ g_enc, FusedGrads G,
                                                            uint32_t* __restrict__ zero_ptr, int zero_words) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // FB_LDS_FLOATS
  tn_zero_words(zero_ptr, zero_words);  // the bucket counters of the table scatter that follows on this stream
  const float* src = pack + PACK_BWD_OFF;
  for (int i = threadIdx.x * 4; i < PACK_BWD_FLOATS; i += blockDim.x * 4) {
    // (split-bf16 head: the two 64-wide head layers' backward fragments -- the same 8192 words -- come from the bf16 region)
    const float* sp = (BF3 && i >= fwd_off(2) && i < fwd_off(4)) ? pack + PACK_BF_BWD_OFF + (i - fwd_off(2)) : src + i;
    *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(sp);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
  float* bufY = lds + PACK_BWD_FLOATS + wv * FB_WAVE_FLOATS;
  float* bufX = bufY + FB_TILE_FLOATS;
  float* bufG = bufX + FB_TILE_FLOATS;             // g3 [32][4]
  int* bufC = reinterpret_cast<int*>(bufG + 128);  // camera of each sample of the tile
  const int64_t ntiles = tn_cdiv(P, TILE);
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
#define AB(layer, t, m, r) lds[fwd_off(layer) + ((((t) * layer_mo(layer) + (m)) * 16 + (r)) << 6) + lane]
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const f32x4v zero4 = {0, 0, 0, 0};
  // weight-gradient accumulators (whole kernel)
  f32x16 acc3[2][2], acc2[2][1], acc0[2][1];
  f32x4v acc4[4], acc1[4];
  float bs3[2] = {0.f, 0.f}, bs2[2] = {0.f, 0.f}, bs0[2] = {0.f, 0.f}, bs4 = 0.f, bs1 = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    acc0[a][0] = zero16;
    acc2[a][0] = zero16;
#pragma unroll
    for (int b = 0; b < 2; ++b) acc3[a][b] = zero16;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) { acc4[b] = zero4; acc1[b] = zero4; }
  // appearance-embedding rows.  The embedding is an INPUT of head layer 0 that is constant over a camera's samples, so its gradient is linear
  // in that layer's pre-activation gradient:  gemb[cam][e] = sum_f hw0[f][31 + e] * (sum over the camera's samples of gy_hh1[.][f]).
  // The inner sum is the layer's bias gradient restricted to the camera -- the lane sums the weight-gradient product forms anyway -- so the
  // chain does not compute d(slots 32..63) at all (32 MFMAs per tile less) and no per-tile column sums are taken: lane = feature (32 h + j)
  // keeps a running (camera, sum), flushed into cam_bias[camera][64] on a camera change; k_field_emb_finish multiplies by hw0 afterwards.
  int emb_cam = -1;
  float emb_sum = 0.0f;
  // A wave walks a CONTIGUOUS slab of tiles: consecutive tiles share rays and cameras, so the running sum is flushed (float atomics
  // into the 128-B row of a camera) about once per camera change.  With tiles dealt round-robin every tile started a new camera and all
  // waves of the chip were adding into the rows of the same two cameras at any time: same-line atomics execute one after the other (~25 ns
  // each) -- 170 us of a 320 us kernel.
  const int64_t tpw = tn_cdiv(ntiles, nwaves);
  const int64_t tile_begin = wave * tpw;
  const int64_t tile_end = (wave + 1) * tpw < ntiles ? (wave + 1) * tpw : ntiles;
  // One wave per SIMD: nothing hides a load but the wave's own MFMA work, so every activation tile is requested one phase (>= 32 MFMAs)
  // before its first use, and the first tile of the next iteration during the last phase of this one.
  f32x16 nx0 = zero16, nx1 = zero16;  // next tile's first activation tile (hh2, or h1 for the density-only backward)
  FbSmall nsm;
  if (tile_begin < tile_end) {
    nsm = fb_load_small<DENS_ONLY>(tile_begin, tile_begin, j, h, P, S, C, num_images, hins, d_density, sel, ys, d_rgb, cam_idx);
    nx0 = load_frag(DENS_ONLY ? h1s : hh2s, tile_begin, 2, 0, lane);
    nx1 = load_frag(DENS_ONLY ? h1s : hh2s, tile_begin, 2, 1, lane);
  }
  // FB_TOUCH (experiment, off): L2 warm-up of the NEXT tile's activations -- at the top of a tile every lane reads ONE word of a different
  // 128-byte line of the next tile's hh1 / h1 / (base outputs | encoding), consumed only at the end of the tile, so that the real loads of the
  // next tile find their lines in L2.  Measured 127 -> 132 us: vmcnt retires in order, so this tile's (L2-hit) loads now queue behind the cold
  // warm-up loads of the next one -- the latency moves, it does not go away (profiles/r05_experiments.md).
  float touch_acc = 0.0f;
  for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
    const int64_t p = tile * TILE + j;
    const bool valid = p < P;
    const int64_t pc = valid ? p : P - 1;
    const FbSmall sm = nsm;
    float tch0 = 0.0f, tch1 = 0.0f, tch2 = 0.0f;
    if (FB_TOUCH && tile + 1 < tile_end) {
      const int64_t tn = tile + 1;
      if (!DENS_ONLY) tch0 = hh1s[(tn * 2 * 4 * 64) * 4 + lane * 32];  // 8 KB per tile = 64 lines
      tch1 = h1s[(tn * 2 * 4 * 64) * 4 + lane * 32];
      if (lane < 16) {
        if (!DENS_ONLY) tch2 = hins[(tn * 128) * 4 + lane * 32];       // 2 KB per tile = 16 lines
      } else if (lane < 48) {                                            // level-major encoding: 256 B per level and tile
        const int l = (lane - 16) >> 1;
        if (l < L) tch2 = encs[((int64_t)l * PT + tn * TILE) * 2 + ((lane - 16) & 1) * 32];
      }
    }
    f32x16 di0 = zero16;
    f32x16 s0, s1;  // h1 tile (ReLU mask of the base MLP)
    if (!DENS_ONLY) {
      if (h == 0) {
        *reinterpret_cast<float4*>(bufG + j * 4) = make_float4(sm.g3[0], sm.g3[1], sm.g3[2], sm.g3[3]);
        bufC[j] = sm.cam;
      }
      // ---- d hh2 = hw2^T . g3   (k-steps r=0..3 carry rows R(r,h): 0..3 for h=0, 4..7 (zero padding) for h=1)
      f32x16 dd0 = zero16, dd1 = zero16;
#pragma unroll
      for (int r = 0; r < 4; ++r) { dd0 = MFMA(AB(4, 0, 0, r), sm.g3[r], dd0); dd1 = MFMA(AB(4, 1, 0, r), sm.g3[r], dd1); }
      f32x16 t0 = load_frag(hh1s, tile, 2, 0, lane), t1 = load_frag(hh1s, tile, 2, 1, lane);  // used after the next chain step
#pragma unroll
      for (int r = 0; r < 16; ++r) { dd0[r] = nx0[r] > 0.0f ? dd0[r] : 0.0f; dd1[r] = nx1[r] > 0.0f ? dd1[r] : 0.0f; }
      lds_put_tile(bufX, j, h, 0, nx0); lds_put_tile(bufX, j, h, 1, nx1);  // X of head layer 2: hh2
      lds_put_tile(bufY, j, h, 0, dd0); lds_put_tile(bufY, j, h, 1, dd1);  // dY of head layer 1: gy_hh2
      WAVE_LDS_SYNC();
      if (!(FB_ABLATE & 4)) wgrad_tile16(bufG, 4, C, bufX, lane, acc4, bs4);  // d hw2 += g3^T hh2
      FB_SB();
      // ---- d hh1 = hw1^T . d hh2
      f32x16 dc0 = zero16, dc1 = zero16;
      if (BF3) {
        bf3_tile_fwd<0>(lds + fwd_off(2) + PACK_BF_LAYER_WORDS, lane, dd0, dc0, dc1);
        bf3_tile_fwd<1>(lds + fwd_off(2) + PACK_BF_LAYER_WORDS, lane, dd1, dc0, dc1);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { dc0 = MFMA(AB(3, 0, 0, r), dd0[r], dc0); dc1 = MFMA(AB(3, 1, 0, r), dd0[r], dc1); }
#pragma unroll
        for (int r = 0; r < 16; ++r) { dc0 = MFMA(AB(3, 0, 1, r), dd1[r], dc0); dc1 = MFMA(AB(3, 1, 1, r), dd1[r], dc1); }
      }
      FB_SB();
      WAVE_LDS_SYNC();  // the reads of bufX (hh2) are done
#pragma unroll
      for (int r = 0; r < 16; ++r) { dc0[r] = t0[r] > 0.0f ? dc0[r] : 0.0f; dc1[r] = t1[r] > 0.0f ? dc1[r] : 0.0f; }
      lds_put_tile(bufX, j, h, 0, t0); lds_put_tile(bufX, j, h, 1, t1);    // X of head layer 1: hh1
      // head-input slots 0..31 (X of head layer 0), rebuilt as the forward built them: sh | the 16 saved base outputs.  (Slots 32..63, the
      // camera's embedding row, are not needed: their weight-gradient columns come out of the per-camera sums, emb_finish_jobs.)
      // Only the base outputs stream from HBM: they are requested here, one weight-gradient block ahead; the per-ray piece is a cache hit
      // and is fetched behind that block (8 registers that do not stay live across it)
      const v4f_t* bp = reinterpret_cast<const v4f_t*>(hins) + tile * 128 + lane;
      const v4f_t hb0 = FRAG_NT ? __builtin_nontemporal_load(bp) : bp[0], hb1 = FRAG_NT ? __builtin_nontemporal_load(bp + 64) : bp[64];
      WAVE_LDS_SYNC();
      if (!(FB_ABLATE & 2)) {  // d hw1 += gy_hh2^T hh1
        if (BF3) wgrad_tile32_bf3<2, 2>(bufY, bufX, j, h, acc3, bs3); else wgrad_tile32<2, 2>(bufY, bufX, j, h, acc3, bs3);
      }
      FB_SB();
      WAVE_LDS_SYNC();
      {
        const float* shp = shtab + (int64_t)sm.ray * 16 + h * 8;
        const float4 sa = *reinterpret_cast<const float4*>(shp), sb = *reinterpret_cast<const float4*>(shp + 4);
        lds_put_tile(bufY, j, h, 0, dc0); lds_put_tile(bufY, j, h, 1, dc1);  // dY of head layer 0: gy_hh1
        t0[0] = sa.x; t0[1] = sa.y; t0[2] = sa.z; t0[3] = sa.w; t0[4] = sb.x; t0[5] = sb.y; t0[6] = sb.z; t0[7] = sb.w;
        t0[8] = hb0.x; t0[9] = hb0.y; t0[10] = hb0.z; t0[11] = hb0.w; t0[12] = hb1.x; t0[13] = hb1.y; t0[14] = hb1.z; t0[15] = hb1.w;
      }
      lds_put_tile(bufX, j, h, 0, t0);                                     // X of head layer 0: head input slots 0..31
      s0 = load_frag(h1s, tile, 2, 0, lane); s1 = load_frag(h1s, tile, 2, 1, lane);   // used two blocks further down
      WAVE_LDS_SYNC();
      float tb[2] = {0.0f, 0.0f};  // this tile's share of the layer's bias gradient (lane (j, h): k-parity h of output 32a + j)
      if (!(FB_ABLATE & 2)) {  // d hw0 (slots 0..31) += gy_hh1^T hin
        if (BF3) wgrad_tile32_bf3<2, 1>(bufY, bufX, j, h, acc2, tb); else wgrad_tile32<2, 1>(bufY, bufX, j, h, acc2, tb);
      }
      bs2[0] += tb[0]; bs2[1] += tb[1];
      FB_SB();
      // ---- appearance-embedding rows (see emb_cam above): the tile's bias sums go to the running sums of its camera
      if (!(FB_ABLATE & 1)) {
        const float tt0 = tb[0] + __shfl_xor(tb[0], 32, 64), tt1 = tb[1] + __shfl_xor(tb[1], 32, 64);  // all 32 samples of outputs j, 32 + j
        const int cam0 = __builtin_amdgcn_readfirstlane(sm.cam);  // (lane 0's sample exists: tile * 32 < P)
        if (__ballot(sm.cam >= 0 && sm.cam != cam0) == 0) {
          if (cam0 != emb_cam) {
            if (emb_cam >= 0 && emb_sum != 0.0f) atomicAdd(cam_bias + (int64_t)emb_cam * 64 + lane, emb_sum);
            emb_cam = cam0;
            emb_sum = 0.0f;
          }
          emb_sum += h ? tt1 : tt0;
        } else {
          // a camera boundary inside the tile (ray counts that are not a multiple of the tile): the tile's rows sample by sample; bufY still
          // holds gy_hh1 [32 samples][64] and bufC the samples' cameras (-1 beyond the last sample)
          for (int q = 0; q < 32; ++q) {
            const int c = bufC[q];
            if (c >= 0) {
              if (c != emb_cam) {
                if (emb_cam >= 0 && emb_sum != 0.0f) atomicAdd(cam_bias + (int64_t)emb_cam * 64 + lane, emb_sum);
                emb_cam = c;
                emb_sum = 0.0f;
              }
              emb_sum += bufY[q * TSTR + lane];
            }
          }
        }
      }
      FB_SB();
      // ---- d head-input slots 0..31 = Wslot^T . d hh1 (sh | base outputs: only the base outputs' rows are used below; slots 32..63, the
      // embedding, take their gradient through cam_bias)
      if (BF3) {
        bf3_tile_one<0>(lds + fwd_off(2), lane, dc0, di0);
        bf3_tile_one<1>(lds + fwd_off(2), lane, dc1, di0);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) di0 = MFMA(AB(2, 0, 0, r), dc0[r], di0);
#pragma unroll
        for (int r = 0; r < 16; ++r) di0 = MFMA(AB(2, 0, 1, r), dc1[r], di0);
      }
      FB_SB();
    } else {
      s0 = nx0; s1 = nx1;
    }
    // ---- d base_out rows: slots 16..31 = registers 8..15 of tile 0; row 0 (half 0, reg 0) takes the trunc_exp gradient instead
    float dbo[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) dbo[r] = di0[8 + r];
    {
      float g = sm.g_dens * expf(fminf(fmaxf(sm.pre_logit, -15.0f), 15.0f)) * sm.sel_p;
      if (h == 0) dbo[0] = valid ? g : 0.0f;
    }
    WAVE_LDS_SYNC();
    // dY of base layer 1: gy_bo [32][16] in the first 16 columns of bufY
    *reinterpret_cast<float4*>(bufY + j * TSTR + 4 * h) = make_float4(dbo[0], dbo[1], dbo[2], dbo[3]);
    *reinterpret_cast<float4*>(bufY + j * TSTR + 8 + 4 * h) = make_float4(dbo[4], dbo[5], dbo[6], dbo[7]);
    lds_put_tile(bufX, j, h, 0, s0); lds_put_tile(bufX, j, h, 1, s1);      // X of base layer 1: h1
    f32x16 e0 = load_enc_lm(encs, PT, pc, h, L);                                // used after the next weight-gradient block + chain step
    WAVE_LDS_SYNC();
    if (!(FB_ABLATE & 4)) wgrad_tile16(bufY, TSTR, 16, bufX, lane, acc1, bs1);  // d w1 += gy_bo^T h1
    FB_SB();
    // ---- d h1 = w1^T . d base_out   (k-steps r<8: rows < 16)
    f32x16 dh0 = zero16, dh1 = zero16;
#pragma unroll
    for (int r = 0; r < 8; ++r) { dh0 = MFMA(AB(1, 0, 0, r), dbo[r], dh0); dh1 = MFMA(AB(1, 1, 0, r), dbo[r], dh1); }
#pragma unroll
    for (int r = 0; r < 16; ++r) { dh0[r] = s0[r] > 0.0f ? dh0[r] : 0.0f; dh1[r] = s1[r] > 0.0f ? dh1[r] : 0.0f; }
    FB_SB();
    WAVE_LDS_SYNC();
    lds_put_tile(bufY, j, h, 0, dh0); lds_put_tile(bufY, j, h, 1, dh1);    // dY of base layer 0: gy_h1
    lds_put_tile(bufX, j, h, 0, e0);                                       // X of base layer 0: enc
    {
      // the next tile's first loads ride behind the last 64 MFMAs of this one (clamped: the last iteration re-reads its own tile)
      const int64_t tn = tile + 1 < tile_end ? tile + 1 : tile;
      nsm = fb_load_small<DENS_ONLY>(tn, tn, j, h, P, S, C, num_images, hins, d_density, sel, ys, d_rgb, cam_idx);
      nx0 = load_frag(DENS_ONLY ? h1s : hh2s, tn, 2, 0, lane);
      nx1 = load_frag(DENS_ONLY ? h1s : hh2s, tn, 2, 1, lane);
    }
    WAVE_LDS_SYNC();
    if (!(FB_ABLATE & 2)) wgrad_tile32<2, 1>(bufY, bufX, j, h, acc0, bs0);  // d w0 += gy_h1^T enc
    FB_SB();
    // ---- d enc = w0^T . d h1
    f32x16 de = zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) de = MFMA(AB(0, 0, 0, r), dh0[r], de);
#pragma unroll
    for (int r = 0; r < 16; ++r) de = MFMA(AB(0, 0, 1, r), dh1[r], de);
    if (valid) {  // d enc, level-major [16][P] float2: registers 4g..4g+3 = levels 4g + 2h (f0, f1), 4g + 2h + 1 (f0, f1)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float2*>(g_enc + (int64_t)(4 * g + 2 * h) * 2 * P + 2 * p) = make_float2(de[4 * g], de[4 * g + 1]);
        *reinterpret_cast<float2*>(g_enc + (int64_t)(4 * g + 2 * h + 1) * 2 * P + 2 * p) = make_float2(de[4 * g + 2], de[4 * g + 3]);
      }
    }
    WAVE_LDS_SYNC();
    touch_acc += (tch0 + tch1) + tch2;
  }
#undef AB
  if (touch_acc == 1.2345678e-30f) cam_bias[0] = touch_acc;  // (never: keeps the warm-up loads alive)
  if (FB_ABLATE & 8) return;
  // ---- block sum of the accumulators (plain LDS read-modify-write, one wave per turn: ds_add_f32 is lane-serialised on gfx950), then one
  // burst of global float atomics per block; exact zeros stay exact zeros (v != 0 guard)
  __syncthreads();  // every wave is done with its tile buffers: the sums go there
  float* red = lds + PACK_BWD_FLOATS;
  if (!DENS_ONLY) {
    // the four waves' running per-camera sums: merged per camera inside the block first (a block's slab of rays usually holds one or two
    // cameras), so that the end of the kernel is not 2048 half-waves adding into the same few 128-B rows at once
    float* esum = red + FB_RED_TOTAL;                          // [4 waves][64 sums]
    int* ecam = reinterpret_cast<int*>(esum + 4 * 64);          // [4]
    esum[wv * 64 + lane] = emb_sum;
    if (lane == 0) ecam[wv] = emb_cam;
    __syncthreads();
    if (threadIdx.x < 64) {
      for (int e = 0; e < 4; ++e) {
        const int cam = ecam[e];
        if (cam < 0) continue;
        bool first = true;
        for (int f = 0; f < e; ++f) first = first && (ecam[f] != cam);
        if (!first) continue;
        float tot = 0.0f;
        for (int f = e; f < 4; ++f)
          if (ecam[f] == cam) tot += esum[f * 64 + threadIdx.x];
        if (tot != 0.0f) atomicAdd(cam_bias + (int64_t)cam * 64 + threadIdx.x, tot);
      }
    }
  }
  const int c16 = lane & 15, k16 = lane >> 4;
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#define RED_PUT(idx, v) { float* d_ = &red[idx]; *d_ = (w == 0) ? (v) : *d_ + (v); }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = 32 * a + RROW(r, h);
          if (!DENS_ONLY) {
#pragma unroll
            for (int b = 0; b < 2; ++b) RED_PUT(FB_RED_W3 + o * 64 + 32 * b + j, acc3[a][b][r]);
            RED_PUT(FB_RED_W2 + o * 32 + j, acc2[a][0][r]);  // [64 out][32 slots]
          }
          RED_PUT(FB_RED_W0 + o * 32 + j, acc0[a][0][r]);
        }
        // bias sums: lane (j, h) holds its k-parity share of output 32a + j
        const float s3 = bs3[a] + __shfl_xor(bs3[a], 32, 64), s2 = bs2[a] + __shfl_xor(bs2[a], 32, 64), s0_ = bs0[a] + __shfl_xor(bs0[a], 32, 64);
        if (h == 0) {
          if (!DENS_ONLY) { RED_PUT(FB_RED_B3 + 32 * a + j, s3); RED_PUT(FB_RED_B2 + 32 * a + j, s2); }
          RED_PUT(FB_RED_B0 + 32 * a + j, s0_);
        }
      }
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (!DENS_ONLY) RED_PUT(FB_RED_W4 + (4 * k16 + r) * 64 + 16 * b + c16, acc4[b][r]);
          RED_PUT(FB_RED_W1 + (4 * k16 + r) * 64 + 16 * b + c16, acc1[b][r]);
        }
      float s4 = bs4 + __shfl_xor(bs4, 16, 64); s4 += __shfl_xor(s4, 32, 64);
      float s1_ = bs1 + __shfl_xor(bs1, 16, 64); s1_ += __shfl_xor(s1_, 32, 64);
      if (lane < 16) {
        if (!DENS_ONLY) RED_PUT(FB_RED_B4 + lane, s4);
        RED_PUT(FB_RED_B1 + lane, s1_);
      }
#undef RED_PUT
    }
    __syncthreads();
  }
  for (int t = threadIdx.x; t < FB_RED_TOTAL; t += blockDim.x) {
    const float v = red[t];
    if (v == 0.0f) continue;
    float* dst = nullptr;
    if (t < FB_RED_W2) { if (!DENS_ONLY) dst = G.ghw1 + t; }
    else if (t < FB_RED_W0) {  // d hw0, slots 0..31 (sh | base outputs): [64 out][32 slots] at the head of the region, the rest unused
      const int q = t - FB_RED_W2;
      if (q >= 64 * 32) continue;
      const int col = slot_to_col(q & 31);
      if (!DENS_ONLY && col >= 0) dst = G.ghw0 + (q >> 5) * 63 + col;
    }
    else if (t < FB_RED_W4) dst = G.gw0 + (t - FB_RED_W0);
    else if (t < FB_RED_W1) { const int q = t - FB_RED_W4; if (!DENS_ONLY && (q >> 6) < C) dst = G.ghw2 + q; }
    else if (t < FB_RED_B3) dst = G.gw1 + (t - FB_RED_W1);
    else if (t < FB_RED_B2) { if (!DENS_ONLY) dst = G.ghb1 + (t - FB_RED_B3); }
    else if (t < FB_RED_B0) { if (!DENS_ONLY) dst = G.ghb0 + (t - FB_RED_B2); }
    else if (t < FB_RED_B4) dst = G.gb0 + (t - FB_RED_B0);
    else if (t < FB_RED_B1) { if (!DENS_ONLY && (t - FB_RED_B4) < C) dst = G.ghb2 + (t - FB_RED_B4); }
    else dst = G.gb1 + (t - FB_RED_B1);
    if (dst) atomicAdd(dst, v);
  }
}

// (k_field_bwd_pair -- the same backward with TWO waves per SIMD, each wave of a pair owning half of the output features -- was measured at
// 136-151 us against this kernel's 126 and removed in round 6; what it taught is in profiles/r05_experiments.md.)

// ---- host entry points -----------------------------------------------------------------------------------------------------
static int check_field(const TnField* f, const char* who, bool need_grad) {
  TN_REQUIRE(f != nullptr, "%s: null field", who);
  TN_REQUIRE(f->grid.table && f->w0 && f->b0 && f->w1 && f->b1 && f->hw0 && f->hb0 && f->hw1 && f->hb1 && f->hw2 && f->hb2 && f->emb,
             "%s: null parameter pointer", who);
  TN_REQUIRE(f->grid.num_levels >= 1 && f->grid.num_levels <= TN_MAX_LEVELS, "%s: num_levels %d out of range", who, f->grid.num_levels);
  TN_REQUIRE(f->grid.log2_hashmap_size >= 1 && f->grid.log2_hashmap_size <= 24, "%s: bad log2_hashmap_size", who);
  TN_REQUIRE(f->num_channels >= 1 && f->num_channels <= 4, "%s: num_channels %d unsupported", who, f->num_channels);
  TN_REQUIRE(f->num_images >= 1, "%s: num_images must be >= 1", who);
  if (need_grad)
    TN_REQUIRE(f->grid.table_grad && f->gw0 && f->gb0 && f->gw1 && f->gb1 && f->ghw0 && f->ghb0 && f->ghw1 && f->ghb1 && f->ghw2 && f->ghb2 && f->gemb,
               "%s: null gradient pointer", who);
  return TN_OK;
}
static FieldK make_fieldk(const TnField* f) {
  FieldK k{f->w0, f->b0, f->w1, f->b1, f->hw0, f->hb0, f->hw1, f->hb1, f->hw2, f->hb2, f->emb, f->num_channels, f->num_images};
  return k;
}
// TN_HEAD_BF16X3=1 (opt-in, never the default): the colour head's two 64-wide layers on split-bf16 matrix instructions.  Read per call, so a
// process can compare both paths.
static bool tn_head_bf16x3() {
  const char* e = getenv("TN_HEAD_BF16X3");
  return e != nullptr && e[0] == '1';
}
static int mlp_grid(int64_t P) {
  int64_t tiles = tn_cdiv(P, TILE);
  return (int)std::max<int64_t>(1, std::min<int64_t>(tn_cdiv(tiles, 4), 512));
}

extern "C" int tn_field_pack_weights(const TnField* field, void* workspace, tn_stream_t stream) {
  int rc = check_field(field, "tn_field_pack_weights", false);
  if (rc) return rc;
  TN_REQUIRE(workspace != nullptr, "tn_field_pack_weights: null workspace");
  FieldWs ws = ws_layout(workspace, 0, 0);
  hipLaunchKernelGGL(k_field_pack, dim3((unsigned)pack_blocks(field->num_images)), dim3(256), 0, tn_s(stream), make_fieldk(field), ws.pack);
  TN_CHECK_LAUNCH("tn_field_pack_weights");
  return TN_OK;
}

// the encode's work plan, for tests and diagnostics: out[x][i] = {level, first chunk, chunk count} of item i of XCD x (level -1: unused);
// returns the number of chunks per level (a chunk = 128 * ENC_PER samples)
extern "C" int32_t tn_field_encode_plan(const TnGrid* grid, int64_t num_points, int32_t* out) {
  if (grid == nullptr || out == nullptr || num_points <= 0 || grid->num_levels < 1 || grid->num_levels > TN_MAX_LEVELS) return TN_EINVAL;
  const EncSched sc = make_enc_sched(*grid, num_points);
  for (int x = 0; x < 8; ++x)
    for (int i = 0; i < ENC_MAX_ITEMS; ++i) {
      int32_t* o = out + (x * ENC_MAX_ITEMS + i) * 3;
      o[0] = sc.level[x][i]; o[1] = sc.c0[x][i]; o[2] = sc.first[x][i + 1] - sc.first[x][i];
    }
  return (int32_t)tn_cdiv(num_points, ENC_CHUNK);
}

// positions (one launch) + the XCD-affine encode (one launch); a training workspace also receives d enc / d offset for the backward
static FieldK make_fieldk(const TnField* f);
struct FieldPrepExtra { bool pack; void* zero; int64_t zero_bytes; };
static int launch_encode(const TnField* field, const float* origins, const float* directions, const float* e_bins, int64_t N, int32_t S,
                         const FieldWs& ws, tn_stream_t stream, FieldPrepExtra ex = FieldPrepExtra{false, nullptr, 0}) {
  const int64_t P = N * (int64_t)S;
  const unsigned pb = (unsigned)std::min<int64_t>(tn_cdiv(P, 256), 256 * 16);
  float4* z = reinterpret_cast<float4*>(ex.zero);
  float4* cb = reinterpret_cast<float4*>(ws.cam_bias);  // (training workspaces only)
  const int cbn4 = ws.cam_bias ? std::min(field->num_images, FIELD_MAX_IMAGES) * 16 : 0;
  const int64_t zn4 = ex.zero ? ex.zero_bytes / 16 : 0;
  if (ex.pack)
    hipLaunchKernelGGL(k_field_prep<true>, dim3(pb + pack_blocks(field->num_images)), dim3(256), 0, tn_s(stream), make_fieldk(field), ws.pack, origins, directions, e_bins, N, S,
                       reinterpret_cast<float4*>(ws.pos), ws.sel, ws.sh, z, zn4, cb, cbn4);
  else
    hipLaunchKernelGGL(k_field_prep<false>, dim3(pb), dim3(256), 0, tn_s(stream), make_fieldk(field), ws.pack, origins, directions, e_bins, N, S,
                       reinterpret_cast<float4*>(ws.pos), ws.sel, ws.sh, z, zn4, cb, cbn4);
  TN_CHECK_LAUNCH("tn_field_fwd(positions)");
  const EncSched sc = make_enc_sched(field->grid, P);
  const unsigned grid = 8u * (unsigned)sc.blocks_per_xcd;
  if (ws.jac != nullptr)
    hipLaunchKernelGGL(k_field_encode_xcd<true>, dim3(grid), dim3(256), 0, tn_s(stream), make_gridk(field->grid), sc, reinterpret_cast<const float4*>(ws.pos), P,
                       ws.PT, ws.enc, ws.jac);
  else
    hipLaunchKernelGGL(k_field_encode_xcd<false>, dim3(grid), dim3(256), 0, tn_s(stream), make_gridk(field->grid), sc, reinterpret_cast<const float4*>(ws.pos), P,
                       ws.PT, ws.enc, nullptr);
  TN_CHECK_LAUNCH("tn_field_fwd(encode)");
  return TN_OK;
}

extern "C" int tn_field_fwd(const TnField* field, const float* origins, const float* directions, const int64_t* camera_indices, const float* e_bins,
                            int64_t N, int32_t S, int32_t training, void* workspace, int64_t workspace_bytes, float* density, float* rgb,
                            float* density_pre, tn_stream_t stream) {
  return tn_field_fwd_ex(field, origins, directions, camera_indices, e_bins, N, S, training, workspace, workspace_bytes, density, rgb, density_pre, 0, nullptr, 0,
                         stream);
}
// tn_field_fwd with the weight packing (tn_field_pack_weights) and a zero-fill of `zero` riding in its first launch (csrc/tn_pipeline.hip)
int tn_field_fwd_ex(const TnField* field, const float* origins, const float* directions, const int64_t* camera_indices, const float* e_bins, int64_t N, int32_t S,
                    int32_t training, void* workspace, int64_t workspace_bytes, float* density, float* rgb, float* density_pre, int pack_first, void* zero,
                    int64_t zero_bytes, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  int rc = check_field(field, "tn_field_fwd", false);
  if (rc) return rc;
  TN_REQUIRE(origins && directions && camera_indices && e_bins && workspace && density && rgb, "tn_field_fwd: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_field_fwd: bad N=%lld S=%d", (long long)N, S);
  if (N == 0) return TN_OK;
  int64_t P = N * (int64_t)S;
  FieldWs ws = ws_layout(workspace, P, training);
  TN_REQUIRE((2 * ws.PT + P) * 8 < (1ll << 32), "tn_field_fwd: batch too large for the 32-bit lane offsets of the level-major loads");
  TN_REQUIRE(workspace_bytes >= ws.bytes, "tn_field_fwd: workspace of %lld bytes, tn_field_workspace_bytes(%lld, %d) = %lld", (long long)workspace_bytes,
             (long long)P, training, (long long)ws.bytes);
  TN_REQUIRE(zero == nullptr || (((uintptr_t)zero % 16) == 0 && zero_bytes % 16 == 0), "tn_field_fwd: the buffer to clear must be 16-byte aligned and sized");
  TN_REQUIRE(!training || field->num_images <= FIELD_MAX_IMAGES, "tn_field_fwd: %d cameras, the per-camera tables of the training path are sized for %d",
             field->num_images, FIELD_MAX_IMAGES);
  rc = launch_encode(field, origins, directions, e_bins, N, S, ws, stream, FieldPrepExtra{pack_first != 0, zero, zero_bytes});
  if (rc) return rc;
  // the chain: 2 blocks of FWD_THREADS per CU, each with its own 58.7 KB copy of the packed weights
  const size_t shmem = PACK_FWD_TOTAL * sizeof(float);
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(tn_cdiv(tn_cdiv(P, TILE), FWD_THREADS / 64), 512));
  const int L = field->grid.num_levels;
#define FWD_LAUNCH(TRAIN_, BF3_, ...)                                                                                                          \
  {                                                                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_mlp_fwd<TRAIN_, BF3_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    hipLaunchKernelGGL((k_field_mlp_fwd<TRAIN_, BF3_>), dim3(grid), dim3(FWD_THREADS), shmem, tn_s(stream), ws.pack, ws.enc, L, ws.PT, ws.sel, ws.sh, \
                       camera_indices, field->emb, field->num_images, TRAIN_ ? 1 : 0, N, S, field->num_channels, density, rgb, density_pre, __VA_ARGS__); \
  }
  const bool bf3 = tn_head_bf16x3();
  if (training) {
    if (bf3) FWD_LAUNCH(true, true, ws.h1, ws.hin, ws.hh1, ws.hh2, ws.y) else FWD_LAUNCH(true, false, ws.h1, ws.hin, ws.hh1, ws.hh2, ws.y)
  } else {
    if (bf3) FWD_LAUNCH(false, true, nullptr, nullptr, nullptr, nullptr, nullptr) else FWD_LAUNCH(false, false, nullptr, nullptr, nullptr, nullptr, nullptr)
  }
#undef FWD_LAUNCH
  TN_CHECK_LAUNCH("tn_field_fwd");
  return TN_OK;
}

extern "C" int tn_field_density_fwd(const TnField* field, const float* origins, const float* directions, const float* e_bins, int64_t N, int32_t S,
                                    int32_t training, void* workspace, int64_t workspace_bytes, float* density, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  int rc = check_field(field, "tn_field_density_fwd", false);
  if (rc) return rc;
  TN_REQUIRE(origins && directions && e_bins && workspace && density, "tn_field_density_fwd: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_field_density_fwd: bad N=%lld S=%d", (long long)N, S);
  int64_t P = N * (int64_t)S;
  FieldWs ws = ws_layout(workspace, P, training ? 1 : 0);
  TN_REQUIRE((2 * ws.PT + P) * 8 < (1ll << 32), "tn_field_density_fwd: batch too large for the 32-bit lane offsets of the level-major loads");
  TN_REQUIRE(workspace_bytes >= ws.bytes, "tn_field_density_fwd: workspace of %lld bytes, tn_field_workspace_bytes(%lld, %d) = %lld", (long long)workspace_bytes,
             (long long)P, training ? 1 : 0, (long long)ws.bytes);
  rc = launch_encode(field, origins, directions, e_bins, N, S, ws, stream);
  if (rc) return rc;
  size_t shmem = PACK_FWD_TOTAL * sizeof(float);
  if (training) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_density_only<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k_field_density_only<true>, dim3(mlp_grid(P)), dim3(256), shmem, tn_s(stream), ws.pack, ws.enc, field->grid.num_levels, ws.PT, ws.sel, P, density, ws.h1, ws.hin);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_density_only<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k_field_density_only<false>, dim3(mlp_grid(P)), dim3(256), shmem, tn_s(stream), ws.pack, ws.enc, field->grid.num_levels, ws.PT, ws.sel, P, density, nullptr, nullptr);
  }
  TN_CHECK_LAUNCH("tn_field_density_fwd");
  return TN_OK;
}

// a level range is a grid of its own: table / gradient / resolutions shifted (g_enc columns shift by 2 per level at the call site)
static TnGrid level_range_grid(const TnGrid& g, int level_begin, int level_end) {
  TnGrid sub = g;
  const int64_t T2 = 2ll << g.log2_hashmap_size;  // floats per level
  sub.table = g.table + level_begin * T2;
  sub.table_grad = g.table_grad ? g.table_grad + level_begin * T2 : nullptr;
  sub.num_levels = level_end - level_begin;
  for (int i = 0; i < TN_MAX_LEVELS; ++i) sub.res[i] = (level_begin + i < g.num_levels) ? g.res[level_begin + i] : 0.0f;
  return sub;
}

extern "C" int tn_field_bwd_phase(const TnField* field, const float* origins, const float* directions, const int64_t* camera_indices,
                                  const float* e_bins, const float* d_density, const float* d_rgb, int64_t N, int32_t S, void* workspace,
                                  int64_t workspace_bytes, float* d_origins, float* d_directions, int32_t phases, int32_t level_begin, int32_t level_end,
                                  tn_stream_t stream) {
  return tn_field_bwd_phase_ex(field, origins, directions, camera_indices, e_bins, d_density, d_rgb, N, S, workspace, workspace_bytes, d_origins, d_directions,
                               phases, level_begin, level_end, nullptr, nullptr, stream);
}
int tn_field_bwd_phase_ex(const TnField* field, const float* origins, const float* directions, const int64_t* camera_indices, const float* e_bins,
                          const float* d_density, const float* d_rgb, int64_t N, int32_t S, void* workspace, int64_t workspace_bytes, float* d_origins,
                          float* d_directions, int32_t phases, int32_t level_begin, int32_t level_end, const PoseFinishArgs* fold_cowork,
                          bool* fold_cowork_taken, tn_stream_t stream) {
  if (fold_cowork_taken) *fold_cowork_taken = false;
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  int rc = check_field(field, "tn_field_bwd", true);
  if (rc) return rc;
  TN_REQUIRE(origins && directions && camera_indices && e_bins && d_density && workspace, "tn_field_bwd: null pointer");
  const bool dens_only = d_rgb == nullptr;  // backward of tn_field_density_fwd(training): no colour path at all
  TN_REQUIRE((d_origins == nullptr) == (d_directions == nullptr), "tn_field_bwd: d_origins and d_directions must both be given or both NULL");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_field_bwd: bad N=%lld S=%d", (long long)N, S);
  TN_REQUIRE((phases & ~(TN_BWD_MLP | TN_BWD_SCATTER | TN_BWD_JOIN | TN_BWD_SCATTER_BIN | TN_BWD_SCATTER_FOLD | TN_BWD_FORK_DPOS | TN_BWD_COUNTERS_CLEAN)) == 0 && phases != 0,
             "tn_field_bwd: bad phase set %d", phases);
  TN_REQUIRE(!((phases & TN_BWD_SCATTER) && (phases & (TN_BWD_SCATTER_BIN | TN_BWD_SCATTER_FOLD))), "tn_field_bwd: TN_BWD_SCATTER and its two halves exclude each other");
  if (phases & (TN_BWD_SCATTER | TN_BWD_SCATTER_FOLD))
    TN_REQUIRE(level_begin >= 0 && level_begin < level_end && level_end <= field->grid.num_levels, "tn_field_bwd: bad level range [%d, %d)", level_begin,
               level_end);
  int64_t P = N * (int64_t)S;
  FieldWs ws = ws_layout(workspace, P, 1);
  // a short buffer would be a device out-of-bounds write (the scatter scratch alone is ~0.5 GB at 4096 rays): refused here
  TN_REQUIRE(workspace_bytes >= ws.bytes, "tn_field_bwd: workspace of %lld bytes, tn_field_workspace_bytes(%lld, 1) = %lld", (long long)workspace_bytes, (long long)P,
             (long long)ws.bytes);
  TN_REQUIRE((2 * ws.PT + P) * 8 < (1ll << 32), "tn_field_bwd: batch too large for the 32-bit lane offsets of the level-major loads");
  hipStream_t st = tn_s(stream);
  const int C = field->num_channels;
  TN_REQUIRE(dens_only || !(phases & TN_BWD_MLP) || field->num_images <= FIELD_MAX_IMAGES,
             "tn_field_bwd: %d cameras, the per-camera sums of the appearance embedding's gradient are sized for %d", field->num_images, FIELD_MAX_IMAGES);
  DposArgs dpos_args{};
  bool dpos_cowork = false;
  if (phases & TN_BWD_MLP) {
    // chain + every weight gradient in one launch (k_field_bwd_fused)
    const size_t shmem = FB_LDS_FLOATS * sizeof(float);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(tn_cdiv(tn_cdiv(P, TILE), 4), 256));  // one block per CU (one wave per SIMD)
    FusedGrads G{field->gw0, field->gb0, field->gw1, field->gb1, field->ghw0, field->ghb0, field->ghw1, field->ghb1, field->ghw2, field->ghb2, field->gemb};
    uint32_t* zp;
    int zw;
    tn_grid_scatter_counters(field->grid, P, ws.scatter, &zp, &zw);  // the bin pass then starts without a memset launch of its own
    if (dens_only) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_bwd_fused<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      hipLaunchKernelGGL((k_field_bwd_fused<true, false>), dim3(grid), dim3(256), shmem, st, ws.pack, ws.sel, ws.y, d_rgb, d_density, camera_indices, field->num_images, P, S,
                         C, ws.sh, field->emb, ws.cam_bias, field->grid.num_levels, ws.PT, ws.enc, ws.h1, ws.hin, ws.hh1, ws.hh2, ws.g_enc, G, zp, zw);
    } else if (tn_head_bf16x3()) {  // (opt-in: the colour head's products on split bf16; the density path inside the same kernel stays fp32)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_bwd_fused<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      hipLaunchKernelGGL((k_field_bwd_fused<false, true>), dim3(grid), dim3(256), shmem, st, ws.pack, ws.sel, ws.y, d_rgb, d_density, camera_indices, field->num_images, P,
                         S, C, ws.sh, field->emb, ws.cam_bias, field->grid.num_levels, ws.PT, ws.enc, ws.h1, ws.hin, ws.hh1, ws.hh2, ws.g_enc, G, zp, zw);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_bwd_fused<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      hipLaunchKernelGGL((k_field_bwd_fused<false, false>), dim3(grid), dim3(256), shmem, st, ws.pack, ws.sel, ws.y, d_rgb, d_density, camera_indices, field->num_images, P,
                         S, C, ws.sh, field->emb, ws.cam_bias, field->grid.num_levels, ws.PT, ws.enc, ws.h1, ws.hin, ws.hh1, ws.hh2, ws.g_enc, G, zp, zw);
    }
    // the appearance-embedding rows from the per-camera sums the launch above left in the workspace: at the head of k_field_dpos when that
    // launch follows, else a launch of their own
    if (!dens_only && d_origins == nullptr)
      hipLaunchKernelGGL(k_field_emb_finish, dim3(std::min(field->num_images + 8, 1024)), dim3(256), 0, st, ws.cam_bias,
                         reinterpret_cast<uint32_t*>(ws.cam_bias + (int64_t)FIELD_MAX_IMAGES * 64), field->hw0, field->emb, field->gemb, field->ghw0,
                         field->num_images);
    TN_CHECK_LAUNCH("tn_field_bwd(mlp + weight gradients)");
    if (d_origins != nullptr) {
      // d position from the saved d enc / d offset (75 MB read) on the companion stream, beside the table scatter: the bin pass
      // runs without a d-position path of its own (which gathered the 8 x 16 corners of every sample again: 414 MB of the entry point's traffic)
      // Forked to the companion stream only when the caller says other streams are busy anyway (TN_BWD_FORK_DPOS: the proposal networks'
      // backward runs beside this one).  On a step where the main stream is alone, a second active queue costs more than the ~25 us it hides:
      // measured 0.845 (forked) vs 0.815 ms (in line) per non-update step, 1.24 vs 1.39 ms per update step when nothing forks.
      dpos_args = DposArgs{origins, directions, e_bins, ws.g_enc, ws.jac, N, S, field->grid.num_levels, ws.PT, d_origins, d_directions,
                           dens_only ? nullptr : ws.cam_bias, reinterpret_cast<uint32_t*>(ws.cam_bias + (int64_t)FIELD_MAX_IMAGES * 64), field->hw0, field->emb,
                           field->gemb, field->ghw0, field->num_images};
      // ... and when the whole table scatter follows in this call on its segmented path, the pass does not get a launch at all: it runs in extra
      // blocks of the scatter's bin launch (tn_field_dpos.h) -- on iterations with and without the proposal networks' backward beside it
      // (scripts/step_times.py, same box, launch of its own / forked vs co-work: other steps 0.6187-0.6239 -> 0.6115-0.6157 ms; co-work on the
      // update steps too, where the pass used to fork: update steps 0.9008-0.9025 -> 0.8879-0.8927, other steps another 6-9 us less since the
      // fourth queue never becomes active).  TN_DPOS_COWORK=0: a launch of its own, as before.
      const char* cwe = getenv("TN_DPOS_COWORK");
      dpos_cowork = (phases & TN_BWD_SCATTER) && level_begin == 0 && level_end == field->grid.num_levels && !(cwe && cwe[0] == '0') &&
                    tn_grid_scatter_takes_cowork(field->grid, P, ws.scatter);
      if (!dpos_cowork) {
        hipStream_t side = (phases & TN_BWD_FORK_DPOS) ? tn_fork(st) : nullptr;
        const int64_t tiles = tn_cdiv(P, 32);
        hipLaunchKernelGGL(k_field_dpos, dim3((unsigned)std::min<int64_t>(tn_cdiv(tiles, 4), 256 * 8)), dim3(256), 0, side ? side : st, dpos_args);
      }
      TN_CHECK_LAUNCH("tn_field_bwd(d position)");
    }
  }
  // d position comes from k_field_dpos (MLP phase): the scatter never computes it
  float* sc_do = nullptr;
  float* sc_dd = nullptr;
  if (phases & TN_BWD_SCATTER) {
    TnGrid sub = level_range_grid(field->grid, level_begin, level_end);
    // (counters zeroed by the MLP phase: this call runs both, or the caller vouches for it -- and the scatter covers the whole grid in one go)
    const bool cz = (phases & (TN_BWD_MLP | TN_BWD_COUNTERS_CLEAN)) && level_begin == 0 && level_end == field->grid.num_levels;
    // (the fold's co-work rides only where the d position pass does: the whole segmented scatter in this call)
    const PoseFinishArgs* fcw = (fold_cowork != nullptr && dpos_cowork) ? fold_cowork : nullptr;
    rc = tn_grid_scatter_launch(sub, origins, directions, e_bins, ws.g_enc + (int64_t)level_begin * 2 * P, TN_LD_LEVEL_MAJOR, N, S, sc_do, sc_dd, ws.scatter, st,
                                nullptr, cz, dpos_cowork ? &dpos_args : nullptr, fcw);
    if (fold_cowork_taken && rc == TN_OK) *fold_cowork_taken = fcw != nullptr;
  }
  if (phases & (TN_BWD_SCATTER_BIN | TN_BWD_SCATTER_FOLD)) {
    TN_REQUIRE(tn_grid_scatter_is_binned(field->grid, P, ws.scatter), "tn_field_bwd: the two-step scatter needs a record-based path (TN_SCATTER_MODE=2 or 1 -- not 0 --, a scatter workspace, table <= 2^20 slots)");
    if (phases & TN_BWD_SCATTER_BIN)
      rc = tn_grid_scatter_bin(field->grid, origins, directions, e_bins, ws.g_enc, TN_LD_LEVEL_MAJOR, N, S, sc_do, sc_dd, ws.scatter, st,
                               (phases & (TN_BWD_MLP | TN_BWD_COUNTERS_CLEAN)) != 0);
    if (rc == TN_OK && (phases & TN_BWD_SCATTER_FOLD)) rc = tn_grid_scatter_fold(field->grid, P, ws.scatter, level_begin, level_end, st);
  }
  if (phases & TN_BWD_JOIN) tn_join_all(st);
  return rc;
}

extern "C" int tn_field_bwd(const TnField* field, const float* origins, const float* directions, const int64_t* camera_indices, const float* e_bins,
                            const float* d_density, const float* d_rgb, int64_t N, int32_t S, void* workspace, int64_t workspace_bytes, float* d_origins,
                            float* d_directions, tn_stream_t stream) {
  if (field == nullptr) return check_field(field, "tn_field_bwd", true);
  return tn_field_bwd_phase(field, origins, directions, camera_indices, e_bins, d_density, d_rgb, N, S, workspace, workspace_bytes, d_origins, d_directions,
                            TN_BWD_MLP | TN_BWD_SCATTER | TN_BWD_JOIN, 0, field->grid.num_levels, stream);
}

// ---- data-parallel exchange of the coarse levels in dense form (see include/thermal_nerf_hip.h) -----------------------------------
extern "C" int64_t tn_field_dense_count(const TnField* field, int64_t num_points, int32_t level_begin, int32_t level_end) {
  if (field == nullptr || level_begin < 0 || level_begin >= level_end || level_end > field->grid.num_levels) return 0;
  return tn_grid_dense_count(level_range_grid(field->grid, level_begin, level_end), num_points);
}

extern "C" int tn_field_bwd_scatter_dense(const TnField* field, const float* origins, const float* directions, const float* e_bins, int64_t N, int32_t S,
                                          void* workspace, int64_t workspace_bytes, float* d_origins, float* d_directions, int32_t level_begin,
                                          int32_t level_end, float* dense_sum, tn_stream_t stream) {
  if (N == 0) return TN_OK;
  int rc = check_field(field, "tn_field_bwd_scatter_dense", true);
  if (rc) return rc;
  TN_REQUIRE(origins && directions && e_bins && workspace && dense_sum, "tn_field_bwd_scatter_dense: null pointer");
  TN_REQUIRE((d_origins == nullptr) == (d_directions == nullptr), "tn_field_bwd_scatter_dense: d_origins and d_directions must both be given or both NULL");
  TN_REQUIRE(N > 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_field_bwd_scatter_dense: bad N=%lld S=%d", (long long)N, S);
  TN_REQUIRE(level_begin >= 0 && level_begin < level_end && level_end <= field->grid.num_levels, "tn_field_bwd_scatter_dense: bad level range [%d, %d)",
             level_begin, level_end);
  FieldWs ws = ws_layout(workspace, N * (int64_t)S, 1);
  TN_REQUIRE(workspace_bytes >= ws.bytes, "tn_field_bwd_scatter_dense: workspace of %lld bytes, needs %lld", (long long)workspace_bytes, (long long)ws.bytes);
  // (d position was produced by the MLP phase: k_field_dpos)
  return tn_grid_scatter_launch(level_range_grid(field->grid, level_begin, level_end), origins, directions, e_bins,
                                ws.g_enc + (int64_t)level_begin * 2 * N * S, TN_LD_LEVEL_MAJOR, N, S, nullptr, nullptr, ws.scatter, tn_s(stream), dense_sum);
}

extern "C" int tn_field_dense_fold(const TnField* field, int64_t num_points, int32_t level_begin, int32_t level_end, const float* dense_sum,
                                   tn_stream_t stream) {
  int rc = check_field(field, "tn_field_dense_fold", true);
  if (rc) return rc;
  TN_REQUIRE(level_begin >= 0 && level_begin < level_end && level_end <= field->grid.num_levels && num_points > 0, "tn_field_dense_fold: bad arguments");
  return tn_grid_dense_fold(level_range_grid(field->grid, level_begin, level_end), num_points, dense_sum, tn_s(stream));
}
