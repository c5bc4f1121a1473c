// pfr_match.hip — embedding cosine match with running top-K (the candR@K / gallery-match path).
//
// Reference semantics replaced (file:line in /root/reference):
//   engine/controller.py:77-90 (and 143-160)  per query: score all others with similarity_f, argsort descending, top-K
//   configs/dog_fe/fe_dogs_config.py:89-93     similarity_f = (F.cosine_similarity + 1) / 2   (monotone in the cosine)
//   generate_tsv.py:91-125                     query-vs-gallery ranking, top-100 answers
// The reference builds python lists of pairs (≈8.4 µs per scored pair).  Here: rows are
// L2-normalised once (pfr_l2norm_fwd), scores of a query block against a gallery CHUNK are one MFMA GEMM
// (pfr_conv2d_fwd as a plain GEMM, fp32 scores), and this file keeps, per query, the running top-K list across
// chunks: a radix-select / threshold-filter row kernel (one workgroup per query) that only ever sorts the handful of
// scores above the current K-th best.  Ordering: score descending, ties → lower gallery index (the reference's
// argsort is unstable, so its own order under ties is undefined).
#include "pfr_common.h"

#define SEL_LIST 2048   // max entries sorted per query per chunk (current list + new candidates)
#define SEL_CAP 1536    // direct-collect capacity; above it the K-th value of the chunk is found by radix select
#define SEL_EQCAP 1024  // capacity for entries tied with the K-th value

// descending bitonic sort of n (power of two) u64 keys in LDS by a 256-thread block
__device__ void bitonic_desc(unsigned long long* a, int n) {
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long x = a[i], y = a[l];
          const bool desc = ((i & k) == 0);
          if ((x < y) == desc) { a[i] = y; a[l] = x; }
        }
      }
    }
  __syncthreads();
}

// visits every score of a row with 16-byte loads (rows are 16-byte aligned: ld % 4 == 0), 4 loads in flight per thread
template <typename F>
__device__ __forceinline__ void for_each_score(const float* __restrict__ sr, int n, F&& f) {
  const int n4 = n >> 2;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(sr);
  int q = threadIdx.x;
  for (; q + 3 * 256 < n4; q += 4 * 256) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = s4[q + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) f(4 * (q + u * 256) + e, v[u][e]);
  }
  for (; q < n4; q += 256) {
    const f32x4 v = s4[q];
#pragma unroll
    for (int e = 0; e < 4; ++e) f(4 * q + e, v[e]);
  }
  for (int j = 4 * n4 + threadIdx.x; j < n; j += 256) f(j, sr[j]);
}

// scores: [rows][ld] fp32 of this gallery chunk (n valid columns, global gallery index = col0 + column).
// cur: running list [rows][K] of u64 entries (key<<32 | ~idx), sorted descending; cur_n[rows] valid counts.
__global__ __launch_bounds__(256) void rowselect_kernel(const float* __restrict__ scores, int ld, int n, int col0, int K,
                                                        unsigned long long* __restrict__ cur, int* __restrict__ cur_n,
                                                        const int* __restrict__ self_idx, int* __restrict__ flags,
                                                        uint32_t* __restrict__ thrk) {
  __shared__ unsigned long long list[SEL_LIST];
  __shared__ unsigned int hist[2048];
  __shared__ int s_cnt, s_eq, s_bin, s_need;
  const int row = blockIdx.x;
  const float* sr = scores + (size_t)row * ld;
  unsigned long long* cl = cur + (size_t)row * K;
  const int have = cur_n[row];
  const int self = self_idx ? self_idx[row] : -1;
  // threshold: only scores strictly above the current K-th best can enter (later chunks have higher indices → lose ties)
  uint32_t tkey = 0;
  if (have == K) tkey = (uint32_t)(cl[K - 1] >> 32);
  if (threadIdx.x == 0) { s_cnt = 0; s_eq = 0; }
  __syncthreads();
  // Full running list (every chunk but the first ones): ONE pass appends the few scores above the current K-th best; only
  // if more than SEL_CAP pass does the radix path below re-read the row.  List not full yet: everything is a candidate.
  bool collected = false;
  int cnt;
  if (have == K) {
    for_each_score(sr, n, [&](int j, float sc) {
      const uint32_t k = fkey(sc);
      if (k > tkey && col0 + j != self) {
        const int p = atomicAdd(&s_cnt, 1);
        if (p < SEL_CAP) list[p] = ((unsigned long long)k << 32) | (uint32_t)(~(uint32_t)(col0 + j));
      }
    });
    __syncthreads();
    cnt = s_cnt;
    collected = cnt <= SEL_CAP;
  } else {
    cnt = n - ((self >= col0 && self < col0 + n) ? 1 : 0);
  }
  __syncthreads();
  uint32_t lo_key = tkey;      // collect keys > lo_key ...
  bool strict = (have == K);   // ... (or all when the list is not full yet and no radix bound was needed)
  uint32_t eq_key = 0;
  int need_eq = 0;
  // ---- first chunk of a match (no running list yet, everything is a candidate): a PIVOT instead of the radix select.
  // Cosine scores crowd into a few exponent bins, so the three histogram passes below are 65 k same-address LDS atomics per row
  // each (2.65 ms for 10 k rows).  Any pivot that lets between K and SEL_CAP scores through ONE collect pass gives the exact
  // answer (they are sorted and the best K kept).  Two guesses, cheapest first: (1) mean + z * std of every 16th 16-byte group
  // (scores of a row are close to normal; z from the tail fraction max(512, 4K) / n), (2) the order statistic of a sorted
  // 2048-score sample; if neither lands in the window the radix select below decides as before.
  if (cnt > SEL_CAP && have == 0 && K <= 256 && n >= 16384) {
    const int n4s = n >> 2;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(sr);
    const int want = max(512, 4 * K);
    float* redf = reinterpret_cast<float*>(hist);     // [2][256] block reduction scratch
    auto try_pivot = [&](uint32_t pivot) {
      __syncthreads();
      if (threadIdx.x == 0) s_cnt = 0;
      __syncthreads();
      for_each_score(sr, n, [&](int j, float sc) {
        const uint32_t k = fkey(sc);
        if (k > pivot && col0 + j != self) {
          const int p = atomicAdd(&s_cnt, 1);
          if (p < SEL_CAP) list[p] = ((unsigned long long)k << 32) | (uint32_t)(~(uint32_t)(col0 + j));
        }
      });
      __syncthreads();
      const bool ok = s_cnt >= K && s_cnt <= SEL_CAP;
      __syncthreads();
      return ok;
    };
    {   // (1) normal approximation
      float a = 0.f, q = 0.f;
      int m = 0;
      for (int i = threadIdx.x * 16; i < n4s; i += 256 * 16) {
        const f32x4 v = s4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a += v[e]; q = fmaf(v[e], v[e], q); }
        m += 4;
      }
      redf[threadIdx.x] = a; redf[256 + threadIdx.x] = q; redf[512 + threadIdx.x] = (float)m;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
          redf[threadIdx.x] += redf[threadIdx.x + o];
          redf[256 + threadIdx.x] += redf[256 + threadIdx.x + o];
          redf[512 + threadIdx.x] += redf[512 + threadIdx.x + o];
        }
        __syncthreads();
      }
      const float cntf = fmaxf(redf[512], 1.f), mean = redf[0] / cntf;
      const float var = fmaxf(redf[256] / cntf - mean * mean, 0.f);
      // z of the upper-tail fraction f (Abramowitz & Stegun 26.2.23)
      const float f = fminf(0.4f, (float)want / (float)n), t = sqrtf(-2.f * logf(f));
      const float z = t - (2.515517f + 0.802853f * t + 0.010328f * t * t) / (1.f + 1.432788f * t + 0.189269f * t * t + 0.001308f * t * t * t);
      const uint32_t pivot = fkey(mean + z * sqrtf(var));
      if (try_pivot(pivot)) { collected = true; cnt = s_cnt; }
    }
    if (!collected) {   // (2) order statistic of a sorted sample
      const int gstride = max(32, (n4s + 511) / 512);       // sample <= 512 groups = 2048 scores
      const int ngrp = (n4s + gstride - 1) / gstride;
      for (int i = threadIdx.x; i < ngrp; i += 256) {
        const f32x4 v = s4[(size_t)i * gstride];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = 4 * i * gstride + e;
          list[4 * i + e] = (col0 + j == self) ? 0ull : ((unsigned long long)fkey(v[e]) << 32);
        }
      }
      int p2s = 1;
      while (p2s < 4 * ngrp) p2s <<= 1;
      for (int i = 4 * ngrp + threadIdx.x; i < p2s; i += 256) list[i] = 0ull;
      __syncthreads();
      bitonic_desc(list, p2s);
      int rank = (int)((long)want * (4 * ngrp) / n);
      rank = max(4, min(rank, 4 * ngrp - 1));
      const uint32_t pivot = (uint32_t)(list[rank - 1] >> 32);
      if (try_pivot(pivot)) { collected = true; cnt = s_cnt; }
    }
  }
  if (cnt > SEL_CAP) {
    // ---- radix select (11+11+10 bits) of the K-th largest key of this chunk
    int need = K;
    uint32_t prefix = 0;
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
      const int bits = pass == 2 ? 10 : 11;
      for (int i = threadIdx.x; i < 2048; i += 256) hist[i] = 0;
      __syncthreads();
      for_each_score(sr, n, [&](int j, float sc) {
        const uint32_t k = fkey(sc);
        const bool match = pass == 0 ? true : (pass == 1 ? (k >> 21) == prefix : (k >> 10) == prefix);
        if (match && col0 + j != self) atomicAdd(&hist[(k >> shift) & ((1u << bits) - 1)], 1u);
      });
      __syncthreads();
      if (threadIdx.x == 0) {
        int acc = 0, b = (1 << bits) - 1;
        for (; b > 0; --b) {
          if (acc + (int)hist[b] >= need) break;
          acc += hist[b];
        }
        s_bin = b;
        s_need = need - acc;
      }
      __syncthreads();
      prefix = (prefix << bits) | (uint32_t)s_bin;
      need = s_need;
      __syncthreads();
    }
    eq_key = prefix;          // exact K-th largest key of the chunk
    need_eq = need;           // how many entries equal to it are still needed
    lo_key = eq_key;
    strict = true;
  }
  // ---- collect (unless the single pass above already did)
  if (!collected) {
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    for_each_score(sr, n, [&](int j, float sc) {
      if (col0 + j == self) return;
      const uint32_t k = fkey(sc);
      const bool take = strict ? (k > lo_key) : true;
      const unsigned long long e = ((unsigned long long)k << 32) | (uint32_t)(~(uint32_t)(col0 + j));
      if (take) {
        const int p = atomicAdd(&s_cnt, 1);
        if (p < SEL_CAP) list[p] = e;
      } else if (need_eq > 0 && k == eq_key) {
        const int p = atomicAdd(&s_eq, 1);
        if (p < SEL_EQCAP) reinterpret_cast<unsigned long long*>(hist)[p] = e;  // hist (8 KB) reused as the tie buffer
      }
    });
    __syncthreads();
  }
  int total = min(s_cnt, SEL_CAP);
  if (need_eq > 0) {
    const int neq = min(s_eq, SEL_EQCAP);
    if (s_eq > SEL_EQCAP && threadIdx.x == 0 && flags) atomicOr(flags, 1);  // more ties than the buffer holds
    int p2 = 1;
    while (p2 < neq) p2 <<= 1;
    unsigned long long* eqb = reinterpret_cast<unsigned long long*>(hist);
    for (int i = neq + threadIdx.x; i < p2; i += 256) eqb[i] = 0ull;
    __syncthreads();
    bitonic_desc(eqb, p2);  // equal keys → descending ~idx = ascending index
    const int take = min(need_eq, neq);
    for (int i = threadIdx.x; i < take; i += 256)
      if (total + i < SEL_LIST) list[total + i] = eqb[i];
    total = min(SEL_LIST, total + take);
    __syncthreads();
  }
  // ---- append the current list, sort, keep the best K
  for (int i = threadIdx.x; i < have; i += 256)
    if (total + i < SEL_LIST) list[total + i] = cl[i];
  total = min(SEL_LIST, total + have);
  int p2 = 1;
  while (p2 < total) p2 <<= 1;
  for (int i = total + threadIdx.x; i < p2; i += 256) list[i] = 0ull;
  __syncthreads();
  bitonic_desc(list, p2);
  const int keep = min(K, total);
  for (int i = threadIdx.x; i < keep; i += 256) cl[i] = list[i];
  if (threadIdx.x == 0) {
    cur_n[row] = keep;
    thrk[row] = keep == K ? (uint32_t)(list[K - 1] >> 32) : 0u;   // what the fused filter epilogue compares against
  }
}

// merge the candidate list a filter-epilogue GEMM (pfr_match_scores_filter) appended for this query with its running list
__global__ __launch_bounds__(256) void rowmerge_kernel(const unsigned long long* __restrict__ cand, int cap, int K,
                                                       unsigned long long* __restrict__ cur, int* __restrict__ cur_n,
                                                       uint32_t* __restrict__ thrk, int* __restrict__ ccnt,
                                                       int* __restrict__ flags) {
  __shared__ unsigned long long list[SEL_LIST];
  const int row = blockIdx.x;
  int c = ccnt[row];
  if (c == 0) return;
  if (c > cap) {            // more candidates than the buffer holds: the caller must redo this match on the unfused path
    if (threadIdx.x == 0) atomicOr(flags, 2);
    c = cap;
  }
  unsigned long long* cl = cur + (size_t)row * K;
  const int have = cur_n[row];
  for (int i = threadIdx.x; i < c; i += 256) list[i] = cand[(size_t)row * cap + i];
  for (int i = threadIdx.x; i < have; i += 256) list[c + i] = cl[i];
  const int total = c + have;
  int p2 = 1;
  while (p2 < total) p2 <<= 1;
  for (int i = total + threadIdx.x; i < p2; i += 256) list[i] = 0ull;
  __syncthreads();
  bitonic_desc(list, p2);
  const int keep = min(K, total);
  for (int i = threadIdx.x; i < keep; i += 256) cl[i] = list[i];
  if (threadIdx.x == 0) {
    cur_n[row] = keep;
    thrk[row] = keep == K ? (uint32_t)(list[K - 1] >> 32) : 0u;
    ccnt[row] = 0;
  }
}

// the same for the common late-chunk case — candidates + running list <= 256 entries: ONE WAVE per query, the 256 keys in four registers per lane
// (element e = r * 64 + lane), bitonic network through lane shuffles (partner distances < 64) and register pairs (64, 128): no LDS, no
// workgroup barriers (the block kernel above spends its time in 36 barrier-separated stages per query).  Rows it merges leave ccnt = 0, so the
// block kernel launched behind it returns at once for them and handles only the long lists (early segments).
__global__ __launch_bounds__(256) void rowmerge_wave_kernel(const unsigned long long* __restrict__ cand, int cap, int K, int rows,
                                                            unsigned long long* __restrict__ cur, int* __restrict__ cur_n,
                                                            uint32_t* __restrict__ thrk, int* __restrict__ ccnt) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c = ccnt[row], have = cur_n[row];
  if (c == 0 || c > cap || c + have > 256) return;
  unsigned long long* cl = cur + (size_t)row * K;
  unsigned long long v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = r * 64 + lane;
    v[r] = e < c ? cand[(size_t)row * cap + e] : (e < c + have ? cl[e - c] : 0ull);
  }
#pragma unroll
  for (int k = 2; k <= 256; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
        const int rj = j >> 6;   // partner register
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((r & rj) == 0) {
            const int e = r * 64 + lane;
            const unsigned long long a = v[r], b = v[r | rj];
            const bool desc = (e & k) == 0;
            const unsigned long long hi = a > b ? a : b, lo = a > b ? b : a;
            v[r] = desc ? hi : lo;
            v[r | rj] = desc ? lo : hi;
          }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = r * 64 + lane;
          const unsigned long long a = v[r];
          const unsigned long long b = ((unsigned long long)__shfl_xor((unsigned)(a >> 32), j) << 32) | (unsigned)__shfl_xor((unsigned)a, j);
          const bool keepmax = ((e & k) == 0) == ((e & j) == 0);
          v[r] = keepmax ? (a > b ? a : b) : (a > b ? b : a);
        }
      }
    }
  const int total = c + have, keep = min(K, total);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = r * 64 + lane;
    if (e < keep) cl[e] = v[r];
    if (keep == K && e == K - 1) thrk[row] = (uint32_t)(v[r] >> 32);
  }
  if (lane == 0) {
    cur_n[row] = keep;
    if (keep < K) thrk[row] = 0u;
    ccnt[row] = 0;
  }
}

__global__ void topk_unpack_kernel(const unsigned long long* __restrict__ cur, const int* __restrict__ cur_n, int rows, int K,
                                   float* __restrict__ out_scores, int* __restrict__ out_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K) return;
  const int r = i / K, j = i % K;
  if (j < cur_n[r]) {
    const unsigned long long e = cur[i];
    out_scores[i] = fkey_inv((uint32_t)(e >> 32));
    out_idx[i] = (int)(~(uint32_t)e);
  } else {
    out_scores[i] = -INFINITY;
    out_idx[i] = -1;
  }
}

extern "C" long pfr_topk_state_bytes(int rows, int K) { return (long)rows * K * 8 + (long)rows * 4 + 64 + (long)rows * 8; }

// state: pfr_topk_state_bytes(rows, K) bytes, zeroed by pfr_topk_reset before the first chunk
extern "C" int pfr_topk_reset(void* state, int rows, int K, hipStream_t st) {
  PFR_CHECK_ARG(state, "pfr_topk_reset: null state");
  if (hipMemsetAsync(state, 0, (size_t)pfr_topk_state_bytes(rows, K), st) != hipSuccess) {
    pfr_set_error("pfr_topk_reset: hipMemsetAsync failed");
    return PFR_ERR_HIP;
  }
  return PFR_OK;
}

// merge one gallery chunk of fp32 scores [rows][ld] (n valid columns; gallery index of column j = col0 + j)
extern "C" int pfr_topk_update(const float* scores, int rows, int ld, int n, int col0, int K, void* state, const int* self_idx,
                               hipStream_t st) {
  PFR_CHECK_ARG(scores && state && rows > 0 && n > 0, "pfr_topk_update: bad args");
  PFR_CHECK_ARG(K >= 1 && K <= 512, "pfr_topk_update: K must be in [1,512]");
  PFR_CHECK_ARG(ld % 4 == 0 && (reinterpret_cast<size_t>(scores) & 15) == 0, "pfr_topk_update: score rows must be 16-byte aligned (ld %% 4 == 0)");
  const TopkState t = topk_state(state, rows, K);
  hipLaunchKernelGGL(rowselect_kernel, dim3(rows), dim3(256), 0, st, scores, ld, n, col0, K, t.cur, t.cur_n, self_idx, t.flags, t.thrk);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// merge the candidates appended by pfr_match_scores_filter (cand u64 [rows][cap], cap <= 1536)
extern "C" int pfr_topk_merge(const void* cand, int cap, int rows, int K, void* state, hipStream_t st) {
  PFR_CHECK_ARG(cand && state && rows > 0 && K >= 1 && K <= 512 && cap >= 1 && cap + K <= SEL_LIST, "pfr_topk_merge: bad args");
  const TopkState t = topk_state(state, rows, K);
  if (K <= 256)
    hipLaunchKernelGGL(rowmerge_wave_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, (const unsigned long long*)cand, cap, K, rows, t.cur, t.cur_n,
                       t.thrk, t.ccnt);
  hipLaunchKernelGGL(rowmerge_kernel, dim3(rows), dim3(256), 0, st, (const unsigned long long*)cand, cap, K, t.cur, t.cur_n, t.thrk,
                     t.ccnt, t.flags);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
// overflow / tie-buffer flags of the running state (device int: bit 0 tie buffer, bit 1 candidate buffer)
extern "C" int pfr_topk_flags(const void* state, int rows, int K, int* out_host, hipStream_t st) {
  PFR_CHECK_ARG(state && out_host, "pfr_topk_flags: null pointer");
  const TopkState t = topk_state(const_cast<void*>(state), rows, K);
  if (hipMemcpyAsync(out_host, t.flags, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    pfr_set_error("pfr_topk_flags: copy failed");
    return PFR_ERR_HIP;
  }
  return PFR_OK;
}

extern "C" int pfr_topk_finish(const void* state, int rows, int K, float* out_scores, int* out_idx, hipStream_t st) {
  PFR_CHECK_ARG(state && out_scores && out_idx, "pfr_topk_finish: null pointer");
  const unsigned long long* cur = reinterpret_cast<const unsigned long long*>(state);
  const int* cur_n = reinterpret_cast<const int*>(reinterpret_cast<const char*>(state) + (size_t)rows * K * 8);
  hipLaunchKernelGGL(topk_unpack_kernel, dim3((rows * K + 255) / 256), dim3(256), 0, st, cur, cur_n, rows, K, out_scores, out_idx);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// exact fp32 re-scoring of candidate lists: score = <q[r], g[idx]> on the fp32 (normalised) rows, then re-sort and keep K.
// one workgroup per query; a wave computes one dot product at a time.
// With cand_scores (the reduced-precision scores the candidates were selected on, aligned with cand) it also leaves the row's certificate:
//   cert[row][0] = max |fp32 score - selection score| over the row's candidates (how far the selection scores are off, measured)
//   cert[row][1] = (K-th best fp32 score) - (selection score of the LAST candidate): every gallery row outside the list scored at most that
//                  last selection score, so it can belong to the exact top-K only if its own selection error exceeds this gap (+inf when
//                  the list holds every eligible row)
__global__ __launch_bounds__(256) void rescore_kernel(const float* __restrict__ q, const float* __restrict__ g,
                                                      const float* __restrict__ g_scale, int D,
                                                      const int* __restrict__ cand, const float* __restrict__ cand_scores, int KC, int K,
                                                      float* __restrict__ out_scores, int* __restrict__ out_idx, float* __restrict__ cert) {
  __shared__ unsigned long long list[512];
  __shared__ unsigned s_err;
  if (threadIdx.x == 0) s_err = 0u;
  const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* qr = q + (size_t)row * D;
  int p2 = 1;
  while (p2 < KC) p2 <<= 1;
  for (int i = threadIdx.x; i < p2; i += 256) list[i] = 0ull;
  __syncthreads();
  // four candidates per wave and step: their row gathers (2 KB each, anywhere in the gallery) are requested together — one candidate per
  // step was 63 dependent memory round trips per wave (0.99 ms of the 16 ms match).  Clamped indices + a select: no load inside a branch.
  // (The order of the fmaf chain per candidate is the old one: the same scores to the bit.)
  const int nq = D >> 2;                      // 16-byte pieces per row
  for (int c0 = wave * 4; c0 < KC; c0 += 16) {
    int gi[4];
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gi[u] = c0 + u < KC ? cand[(size_t)row * KC + c0 + u] : -1;
      a[u] = 0.f;
    }
    for (int d = lane; d < nq; d += 64) {
      const f32x4 x = reinterpret_cast<const f32x4*>(qr)[d];
      f32x4 y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) y[u] = reinterpret_cast<const f32x4*>(g + (size_t)(gi[u] < 0 ? 0 : gi[u]) * D)[d];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = fmaf(x[0], y[u][0], a[u]); a[u] = fmaf(x[1], y[u][1], a[u]); a[u] = fmaf(x[2], y[u][2], a[u]); a[u] = fmaf(x[3], y[u][3], a[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v = wave_sum(a[u]);
      if (gi[u] >= 0) {
        if (g_scale) v *= g_scale[gi[u]];      // g = the RAW gallery rows, g_scale = 1 / max(|g_i|, eps): no normalised fp32 copy of the gallery
        if (lane == 0) {
          list[c0 + u] = ((unsigned long long)fkey(v) << 32) | (uint32_t)(~(uint32_t)gi[u]);
          if (cand_scores) atomicMax(&s_err, __float_as_uint(fabsf(v - cand_scores[(size_t)row * KC + c0 + u])));   // (>= 0: bit order = value order)
        }
      }
    }
  }
  __syncthreads();
  bitonic_desc(list, p2);
  for (int i = threadIdx.x; i < K; i += 256) {
    const unsigned long long e = list[i];
    if (e != 0ull) {
      out_scores[(size_t)row * K + i] = fkey_inv((uint32_t)(e >> 32));
      out_idx[(size_t)row * K + i] = (int)(~(uint32_t)e);
    } else {
      out_scores[(size_t)row * K + i] = -INFINITY;
      out_idx[(size_t)row * K + i] = -1;
    }
  }
  if (cert && threadIdx.x == 0) {
    const bool full = cand[(size_t)row * KC + KC - 1] >= 0 && list[K - 1] != 0ull;
    cert[2 * row] = __uint_as_float(s_err);
    cert[2 * row + 1] = full ? fkey_inv((uint32_t)(list[K - 1] >> 32)) - cand_scores[(size_t)row * KC + KC - 1] : INFINITY;
  }
}

extern "C" int pfr_topk_rescore_cert(const float* q, const float* g, const float* g_scale, int rows, int D, const int* cand,
                                     const float* cand_scores, int KC, int K, float* out_scores, int* out_idx, float* cert, hipStream_t st) {
  PFR_CHECK_ARG(q && g && cand && cand_scores && cert && out_scores && out_idx, "pfr_topk_rescore_cert: null pointer");
  PFR_CHECK_ARG(D % 4 == 0 && KC <= 512 && K >= 1 && K <= KC, "pfr_topk_rescore_cert: need D %% 4 == 0, 1 <= K <= KC <= 512");
  hipLaunchKernelGGL(rescore_kernel, dim3(rows), dim3(256), 0, st, q, g, g_scale, D, cand, cand_scores, KC, K, out_scores, out_idx, cert);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

extern "C" int pfr_topk_rescore(const float* q, const float* g, const float* g_scale, int rows, int D, const int* cand, int KC, int K,
                                float* out_scores, int* out_idx, hipStream_t st) {
  PFR_CHECK_ARG(q && g && cand && out_scores && out_idx, "pfr_topk_rescore: null pointer");
  PFR_CHECK_ARG(D % 4 == 0 && KC <= 512 && K <= KC, "pfr_topk_rescore: need D %% 4 == 0, K <= KC <= 512");
  hipLaunchKernelGGL(rescore_kernel, dim3(rows), dim3(256), 0, st, q, g, g_scale, D, cand, (const float*)nullptr, KC, K, out_scores, out_idx, (float*)nullptr);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// pair scores of the validation protocol: out[p] = (cos(a_p, b_p) + 1) / 2 with F.cosine_similarity's eps clamp
__global__ __launch_bounds__(256) void pair_sim_kernel(const float* __restrict__ emb, int D, const long* __restrict__ ia,
                                                       const long* __restrict__ ib, int P, float eps, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const float* a = emb + (size_t)ia[p] * D;
  const float* b = emb + (size_t)ib[p] * D;
  float ab = 0.f, aa = 0.f, bb = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float x = a[d], y = b[d];
    ab = fmaf(x, y, ab); aa = fmaf(x, x, aa); bb = fmaf(y, y, bb);
  }
  ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
  if (lane == 0) out[p] = (ab / (fmaxf(sqrtf(aa), eps) * fmaxf(sqrtf(bb), eps)) + 1.f) * 0.5f;
}
extern "C" int pfr_pair_similarity(const float* emb, int D, const long* idx_a, const long* idx_b, int P, float eps, float* out,
                                   hipStream_t st) {
  PFR_CHECK_ARG(emb && idx_a && idx_b && out, "pfr_pair_similarity: null pointer");
  if (P == 0) return PFR_OK;
  hipLaunchKernelGGL(pair_sim_kernel, dim3((P + 3) / 4), dim3(256), 0, st, emb, D, idx_a, idx_b, P, eps, out);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// card centroid of the reference's "mean strategy" (generate_tsv.py:71-78): the card-vs-card score is the mean over the
// photo cross product of (cos+1)/2 = (⟨mean_i q̂_i, mean_j ĝ_j⟩ + 1)/2, so each card is reduced ONCE to the mean of its
// L2-normalised photo embeddings and card matching becomes the same GEMM + top-K as photo matching.
// one wave per card; seg[c] .. seg[c+1] are the photo rows of card c.
template <typename TOo>
__global__ __launch_bounds__(256) void card_centroid_kernel(const float* __restrict__ emb, const long* __restrict__ seg, int ncards,
                                                            int D, float eps, float* __restrict__ cent32, TOo* __restrict__ cent) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= ncards) return;
  const long r0 = seg[c], r1 = seg[c + 1];
  const float invn = r1 > r0 ? 1.f / (float)(r1 - r0) : 0.f;
  for (int d0 = lane; d0 < D; d0 += 64) {
    if (cent32) cent32[(size_t)c * D + d0] = 0.f;
  }
  for (long r = r0; r < r1; ++r) {
    const float* x = emb + (size_t)r * D;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) ss = fmaf(x[d], x[d], ss);
    ss = wave_sum(ss);
    const float inv = invn / fmaxf(sqrtf(ss), eps);
    for (int d = lane; d < D; d += 64) cent32[(size_t)c * D + d] += x[d] * inv;
  }
  if (cent)
    for (int d = lane; d < D; d += 64) cent[(size_t)c * D + d] = from_f32<TOo>(cent32[(size_t)c * D + d]);
}

extern "C" int pfr_card_centroids(const float* emb, const long* seg, int ncards, int D, float eps, float* cent32, void* cent,
                                  int cent_dtype, hipStream_t st) {
  PFR_CHECK_ARG(emb && seg && cent32 && ncards > 0, "pfr_card_centroids: bad args");
  const dim3 grid((ncards + 3) / 4);
  if (cent && cent_dtype == PFR_BF16)
    hipLaunchKernelGGL(card_centroid_kernel<bf16_t>, grid, dim3(256), 0, st, emb, seg, ncards, D, eps, cent32, (bf16_t*)cent);
  else
    hipLaunchKernelGGL(card_centroid_kernel<float>, grid, dim3(256), 0, st, emb, seg, ncards, D, eps, cent32, (float*)cent);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// Head/body score fusion of the inference ranking (generate_tsv.py:91-110).  In: the two centroid dot-product chunks
// (head, body) of a query block against a gallery chunk.  Out (over `head`): the fused card score, −inf where the
// reference `continue`s (different species `type`, or both modality scores 0).
//   s0 = both cards have head vectors ? max((dot_head + 1)/2, 0) : 0        (mean_strategy_cal_scores, :71-78)
//   s1 = both cards have body vectors ? max((dot_body + 1)/2, 0) : 0
//   score = (query has no head vectors || (s0 == 0 && s1 > thr[type-1])) ? s1 : s0          (:107)
// flags byte per card: bit 0 = has head vectors, bit 1 = has body vectors, bits 2..7 = type.
struct FuseThr { float v[8]; };
__global__ __launch_bounds__(256) void card_fuse_kernel(float* __restrict__ head, const float* __restrict__ body, int ld, int n,
                                                        int col0, const unsigned char* __restrict__ qf,
                                                        const unsigned char* __restrict__ gf, FuseThr thr) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int r = blockIdx.y;
  const unsigned q = qf[r], g = gf[col0 + j];
  const size_t o = (size_t)r * ld + j;
  float out = -INFINITY;
  if ((q >> 2) == (g >> 2)) {
    const float s0 = (q & g & 1u) ? fmaxf((head[o] + 1.f) * 0.5f, 0.f) : 0.f;
    const float s1 = (q & g & 2u) ? fmaxf((body[o] + 1.f) * 0.5f, 0.f) : 0.f;
    if (s0 + s1 != 0.f) {
      const unsigned t = (q >> 2) - 1u;
      const float th = thr.v[t < 8u ? t : 7u];
      out = (!(q & 1u) || (s0 == 0.f && s1 > th)) ? s1 : s0;
    }
  }
  head[o] = out;
}

extern "C" int pfr_card_fuse_scores(float* head_scores, const float* body_scores, int rows, int ld, int n, int col0,
                                    const unsigned char* q_flags, const unsigned char* g_flags, const float* thresholds,
                                    int n_types, hipStream_t st) {
  PFR_CHECK_ARG(head_scores && body_scores && q_flags && g_flags && thresholds, "pfr_card_fuse_scores: null pointer");
  PFR_CHECK_ARG(rows > 0 && n > 0 && ld >= n && n_types >= 1 && n_types <= 8 && rows <= 65535, "pfr_card_fuse_scores: bad sizes");
  FuseThr t;
  for (int i = 0; i < 8; ++i) t.v[i] = thresholds[i < n_types ? i : n_types - 1];
  hipLaunchKernelGGL(card_fuse_kernel, dim3((n + 255) / 256, rows), dim3(256), 0, st, head_scores, body_scores, ld, n, col0,
                     q_flags, g_flags, t);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ---- verification-pair curve (engine/controller.py:112-183: what torchmetrics' ROC / AUROC / AveragePrecision / StatScores
// derive from the pair scores): scores sorted descending (ties: lower pair index first), the running count of genuine pairs, and
// the end-of-run flags of equal scores — the sort/scan part of `_evaluate` on the device.  The pair set is small
// (the reference evaluates 20 000 pairs), so ONE workgroup runs the whole bitonic network over 64-bit keys
// (sortable score << 32 | ~index << 1 | label) in an L2-resident global scratch buffer, then a block scan.
__global__ __launch_bounds__(1024) void pair_curve_kernel(const float* __restrict__ scores, const int* __restrict__ labels, int P,
                                                          int Ppad, unsigned long long* __restrict__ keys,
                                                          float* __restrict__ sorted_scores, int* __restrict__ cum_tp,
                                                          unsigned char* __restrict__ run_end) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < Ppad; i += 1024)
    // (+ 0.f: -0.0 and +0.0 are one threshold)
    keys[i] = i < P ? (((unsigned long long)fkey(scores[i] + 0.f) << 32) | ((uint32_t)(~(uint32_t)i) << 1) | (uint32_t)(labels[i] != 0)) : 0ull;
  __syncthreads();
  for (int k = 2; k <= Ppad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < Ppad; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = keys[i], b = keys[l];
          const bool desc = ((i & k) == 0);
          if ((a < b) == desc) { keys[i] = b; keys[l] = a; }
        }
      }
      __syncthreads();
    }
  // block scan of the labels in sorted order: thread t owns the contiguous range [t*per, (t+1)*per)
  const int per = (P + 1023) / 1024;
  const int lo = tid * per, hi = min(P, lo + per);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += (int)(keys[i] & 1ull);
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - s;   // exclusive prefix
  for (int i = lo; i < hi; ++i) {
    const unsigned long long e = keys[i];
    run += (int)(e & 1ull);
    cum_tp[i] = run;
    sorted_scores[i] = fkey_inv((uint32_t)(e >> 32));
    run_end[i] = (i == P - 1 || (uint32_t)(keys[i + 1] >> 32) != (uint32_t)(e >> 32)) ? 1 : 0;
  }
}

extern "C" long pfr_pair_curve_ws_bytes(int P) {
  long n = 1;
  while (n < P) n <<= 1;
  return n * 8;
}

extern "C" int pfr_pair_curve(const float* scores, const int* labels, int P, void* workspace, float* sorted_scores, int* cum_tp,
                              unsigned char* run_end, hipStream_t st) {
  PFR_CHECK_ARG(scores && labels && workspace && sorted_scores && cum_tp && run_end && P > 0 && P <= (1 << 24), "pfr_pair_curve: bad args");
  int n = 1;
  while (n < P) n <<= 1;
  hipLaunchKernelGGL(pair_curve_kernel, dim3(1), dim3(1024), 0, st, scores, labels, P, n, (unsigned long long*)workspace, sorted_scores,
                     cum_tp, run_end);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
