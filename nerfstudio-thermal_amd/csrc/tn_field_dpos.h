// The main field's d position pass (k_field_dpos) and the embedding-gradient finisher that rides at its head, as device functions: tn_field.hip
// launches them as kernels of their own; tn_scatter.hip runs them in EXTRA BLOCKS of the table scatter's bin launch (k_seg_bin) on iterations
// where nothing else runs beside the main stream -- a streaming pass beside a pass that is bound by vector-instruction issue, in one queue.
// `bid` / `nblk` stand in for the block index / block count of a launch of their own.
#pragma once
#include "tn_common.h"

struct DposArgs {
  const float* origins; const float* directions; const float* e_bins; const float* g_enc; const float* jac;
  int64_t N; int S; int L; int64_t PT;
  float* d_origins; float* d_directions;
  float* cam_bias; uint32_t* fin_counter; const float* hw0; const float* emb; float* gemb; float* ghw0; int num_images;  // cam_bias NULL: no finisher
};

// What the per-camera sums cam_bias[cam][f] = sum over the camera's samples of gy_hh1[.][f] (k_field_bwd_fused) stand for.  The appearance
// embedding is an input of head layer 0 that is constant over a camera's samples, so both of its gradients are linear in those sums
// (hw0 = the head's first nn.Linear weight [64][63]; its columns 31..62 multiply the embedding: fields/nerfacto_field.py:288-300):
//   gemb[cam][e]    += sum_f   hw0[f][31 + e] * cam_bias[cam][f]        job `cam` (threads 0..63 of a block)
//   ghw0[f][31 + e] += sum_cam cam_bias[cam][f] * emb[cam][e]           jobs num_images .. num_images + 7 (256 of the 2048 entries each)
// -- the fused kernel therefore neither computes d(head-input slots 32..63) nor the weight-gradient tile of those slots (64 MFMAs per tile less).
// Jobs are dealt to the blocks of the launch round-robin; the last block to finish (one relaxed counter; every block's reads of cam_bias are
// complete before it counts itself in) clears cam_bias and the counter for the next backward; blocks without a job do not take part.  The only writers of gemb and of those columns of
// ghw0 on their stream at this point: plain read-modify-write.  Rides at the head of k_field_dpos when that launch follows, else k_field_emb_finish.
__device__ __forceinline__ void emb_finish_jobs(float* __restrict__ cam_bias, uint32_t* __restrict__ counter, const float* __restrict__ hw0,
                                                const float* __restrict__ emb, float* __restrict__ gemb, float* __restrict__ ghw0, int num_images,
                                                float* sums /* 64 floats of LDS */, int* flag /* 1 int of LDS */, unsigned bid, unsigned nblk) {
  const int t = threadIdx.x;
  const int njobs = num_images + 8;
  if ((int)bid >= njobs) return;  // only the blocks with a job count themselves in (one same-address atomic each: ~25 ns apiece)
  const unsigned workers = (unsigned)njobs < nblk ? (unsigned)njobs : nblk;
  for (int job = bid; job < njobs; job += nblk) {  // (block-uniform trip count)
    if (job < num_images) {
      const int cam = job;
      float v = 0.0f;
      if (t < 64) {
        v = cam_bias[(int64_t)cam * 64 + t];
        sums[t] = v;
      }
      __syncthreads();
      if (t < 64 && __ballot(v != 0.0f) != 0ull) {  // (wave 0 as a whole; a camera without samples in this batch is skipped)
        const int e = t & 31, half = t >> 5;
        float a = 0.0f;
        for (int f = 0; f < 32; ++f) a += hw0[(32 * half + f) * 63 + 31 + e] * sums[32 * half + f];
        a += __shfl_xor(a, 32, 64);
        if (half == 0 && a != 0.0f) gemb[(int64_t)cam * 32 + e] += a;
      }
      __syncthreads();
    } else {
      const int o = (job - num_images) * 256 + t;  // entry (f, e) of the embedding columns of d hw0
      if (t < 256 && o < 2048) {
        const int f = o >> 5, e = o & 31;
        float a = 0.0f;
        for (int cam = 0; cam < num_images; ++cam) {
          const float b = cam_bias[(int64_t)cam * 64 + f];
          if (b != 0.0f) a += b * emb[(int64_t)cam * 32 + e];
        }
        if (a != 0.0f) ghw0[f * 63 + 31 + e] += a;
      }
    }
  }
  __syncthreads();  // every thread of the block has consumed what it read from cam_bias
  if (t == 0) *flag = (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == workers - 1) ? 1 : 0;
  __syncthreads();
  if (*flag) {
    for (int i = t; i < num_images * 64; i += blockDim.x) cam_bias[i] = 0.0f;
    if (t == 0) *counter = 0u;
  }
}

// d position of every sample from d enc and the saved derivatives: dp_axis = sum_l sum_f g_enc[2l + f] * jac[l][f][axis]; then the
// backward of contraction / frustum position and the per-ray sums into d origins / d directions (same arithmetic as the table scatter's
// own d-position path, which stays for tn_hash_scatter and the proposal grids).  lane = (sample j of the tile, half h of the levels).
__device__ __forceinline__ void field_dpos_body(const DposArgs& a, unsigned bid, unsigned nblk, float* emb_sums /* 64 floats of LDS */, int* emb_flag) {
  // (the embedding's gradients of the MLP phase before this launch, from its per-camera sums)
  if (a.cam_bias != nullptr) emb_finish_jobs(a.cam_bias, a.fin_counter, a.hw0, a.emb, a.gemb, a.ghw0, a.num_images, emb_sums, emb_flag, bid, nblk);
  const int64_t N = a.N, PT = a.PT;
  const int S = a.S, L = a.L;
  const int64_t P = N * (int64_t)S;
  const int64_t ntiles = tn_cdiv(P, 32);
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  typedef float v2f_t __attribute__((ext_vector_type(2)));
  for (int64_t tile = (int64_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < ntiles; tile += (int64_t)nblk * (blockDim.x >> 6)) {
    const int64_t p = tile * 32 + j;
    const bool live = p < P;
    const int64_t pc = live ? p : P - 1;
    float dx = 0.f, dy = 0.f, dz = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float2 gl[2];
      v2f_t jl[2][3];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int l = 4 * q + 2 * h + e;
        if (l < L) {
          gl[e] = *reinterpret_cast<const float2*>(a.g_enc + (int64_t)l * 2 * P + 2 * pc);  // level-major [16][P] float2
          const v2f_t* jp = reinterpret_cast<const v2f_t*>(a.jac) + (int64_t)l * 3 * PT + pc;
#pragma unroll
          for (int k = 0; k < 3; ++k) jl[e][k] = __builtin_nontemporal_load(jp + k * PT);
        } else {
          gl[e] = make_float2(0.f, 0.f);
#pragma unroll
          for (int k = 0; k < 3; ++k) jl[e][k] = v2f_t{0.f, 0.f};
        }
      }
      // plane k of a level = (d enc_0 / d o_k, d enc_1 / d o_k) * res
      dx += gl[0].x * jl[0][0].x + gl[0].y * jl[0][0].y + gl[1].x * jl[1][0].x + gl[1].y * jl[1][0].y;
      dy += gl[0].x * jl[0][1].x + gl[0].y * jl[0][1].y + gl[1].x * jl[1][1].x + gl[1].y * jl[1][1].y;
      dz += gl[0].x * jl[0][2].x + gl[0].y * jl[0][2].y + gl[1].x * jl[1][2].x + gl[1].y * jl[1][2].y;
    }
    dx += __shfl_xor(dx, 32, 64); dy += __shfl_xor(dy, 32, 64); dz += __shfl_xor(dz, 32, 64);
    const int64_t ray = tn_div_index(pc, S, P);
    const int s = (int)(pc - ray * S);
    const float* o = a.origins + ray * 3;
    const float* d = a.directions + ray * 3;
    const float* eb = a.e_bins + ray * (S + 1) + s;
    const float st = eb[0], en = eb[1];
    const Contracted c = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], st, en);
    float wx, wy, wz;
    tn_contract_bwd(c, dx, dy, dz, wx, wy, wz);
    if (!live || h != 0) { wx = wy = wz = 0.0f; }
    const float tm = (st + en) / 2.0f;
    float v[6] = {wx, wy, wz, wx * tm, wy * tm, wz * tm};
    // segmented sums over the lanes of one ray (consecutive samples of a ray sit in consecutive lanes of a half-wave); the last lane of a
    // segment adds the segment's sums
    const int r32 = (int)ray;
    const int prev = __shfl_up(r32, 1, 64);
    const bool head = (j == 0) || (prev != r32);
    const unsigned long long H = __ballot(head);
    const int start = 63 - __clzll(H & (~0ull >> (63 - lane)));  // first lane of this lane's segment
    const bool tail = (j == 31) || ((H >> (lane + 1)) & 1ull);
#pragma unroll
    for (int o2 = 1; o2 < 32; o2 <<= 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float t = __shfl_up(v[k], o2, 64);
        if (lane - o2 >= start) v[k] += t;
      }
    }
    // (no `live` test: in a last, partial tile the lanes beyond the last sample belong to the last ray's segment -- pc = P - 1 -- and carry zeros,
    // so that segment's tail IS such a lane; testing `live` here dropped the partial tile's share of the last ray: any N * S % 32 != 0)
    if (h == 0 && tail) {
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (v[k] != 0.0f) atomicAdd((k < 3 ? a.d_origins : a.d_directions) + ray * 3 + (k % 3), v[k]);
    }
  }
}

