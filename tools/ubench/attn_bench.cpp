// Attention micro-benchmark + self-check through the C ABI (no torch): bf16 forward / backward against the library's
// exact-f32 kernels on the same (bf16-rounded) inputs, then HIP-event timings.
// build: hipcc -O2 -std=c++17 tools/ubench/attn_bench.cpp -Iinclude -Lgmlm_amd -lgmlm_hip -Wl,-rpath,'$ORIGIN/../../gmlm_amd' -o tools/ubench/attn_bench
// run:   tools/ubench/attn_bench [ignored] [1 = also backward] [ignored] [case-name filter]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include "gmlm_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define GK(x) do { int r_ = (x); if (r_ != 0) { printf("gmlm error %d (%s) at %s:%d\n", r_, gmlm_last_error(), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

#ifdef GMLM_ATTN_STAMP
extern "C" void gmlm_debug_set_stamp_buffer(void* p);
#endif
struct Case { const char* tag; int64_t b, h, l, d; bool masked; float drop; };

int main(int argc, char** argv) {
  std::vector<int> variants = {1};
  if (argc > 1) { variants.clear(); char* s = strdup(argv[1]); for (char* t = strtok(s, ","); t; t = strtok(nullptr, ",")) variants.push_back(atoi(t)); }
  const bool do_bwd = argc > 2 && atoi(argv[2]);
  if (argc > 3) setenv("GMLM_ATTN_NW", argv[3], 1);
  const char* only = argc > 4 ? argv[4] : nullptr;
  { int cu = 0, ws_ = 0; char arch[64] = {0}; GK(gmlm_device_check(&cu, &ws_, arch, 64)); printf("device %s, %d CUs\n", arch, cu); }
  const Case cases[] = {{"mha_L512", 32, 12, 512, 64, true, 0.f}, {"mha_L128", 256, 12, 128, 64, true, 0.f},
                        {"xattn_N5201", 1, 8, 5201, 96, false, 0.f}, {"xattn_N20804", 1, 8, 20804, 96, false, 0.f},
                        {"mha_L512_drop", 32, 12, 512, 64, true, 0.1f}, {"mha_L77_odd", 64, 12, 77, 64, true, 0.f},
                        {"xattn_N999", 1, 8, 999, 96, false, 0.f},
                        {"mha_L512_b8", 8, 12, 512, 64, true, 0.f}, {"mha_L512_b128", 128, 12, 512, 64, true, 0.f},
                        {"mha_L512_full", 32, 12, 512, 64, false, 0.f}, {"mha_L2048", 8, 12, 2048, 64, false, 0.f},
                        {"mha_L2048_masked", 16, 12, 2048, 64, true, 0.f}, {"mha_L4096_masked", 8, 12, 4096, 64, true, 0.f}};
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (const Case& c : cases) {
    if (only && (only[0] == '=' ? strcmp(c.tag, only + 1) != 0 : !strstr(c.tag, only))) continue;   // "=name": exact match
    const int64_t n = c.b * c.l * c.h * c.d, rows = c.b * c.l;
    std::vector<uint16_t> hq(n), hk(n), hv(n), hgo(n);
    std::vector<float> fq(n), fk(n), fv(n), fgo(n);
    for (int64_t i = 0; i < n; ++i) {
      hq[i] = f2bf(nd(rng)); hk[i] = f2bf(nd(rng)); hv[i] = f2bf(nd(rng)); hgo[i] = f2bf(nd(rng));
      fq[i] = bf2f(hq[i]); fk[i] = bf2f(hk[i]); fv[i] = bf2f(hv[i]); fgo[i] = bf2f(hgo[i]);
    }
    std::vector<int32_t> hlen(c.b);
    for (auto& x : hlen) x = (int32_t)(c.l / 2 + rng() % (c.l - c.l / 2 + 1));
    uint16_t *q, *k, *v, *o, *go, *dq, *dk, *dv; float *qf, *kf, *vf, *of, *gof, *dqf, *dkf, *dvf, *lse, *lsef; int32_t* len = nullptr; void* ws;
    CK(hipMalloc(&q, n * 2)); CK(hipMalloc(&k, n * 2)); CK(hipMalloc(&v, n * 2)); CK(hipMalloc(&o, n * 2)); CK(hipMalloc(&go, n * 2));
    CK(hipMalloc(&dq, n * 2)); CK(hipMalloc(&dk, n * 2)); CK(hipMalloc(&dv, n * 2));
    CK(hipMalloc(&qf, n * 4)); CK(hipMalloc(&kf, n * 4)); CK(hipMalloc(&vf, n * 4)); CK(hipMalloc(&of, n * 4)); CK(hipMalloc(&gof, n * 4));
    CK(hipMalloc(&dqf, n * 4)); CK(hipMalloc(&dkf, n * 4)); CK(hipMalloc(&dvf, n * 4));
    CK(hipMalloc(&lse, rows * c.h * 4)); CK(hipMalloc(&lsef, rows * c.h * 4));
    const size_t wsb = gmlm_attention_bwd_workspace_bytes(c.b, c.h, c.l, c.l, c.d);
    CK(hipMalloc(&ws, wsb));
    CK(hipMemcpy(q, hq.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(k, hk.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hv.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(go, hgo.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(qf, fq.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(kf, fk.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vf, fv.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(gof, fgo.data(), n * 4, hipMemcpyHostToDevice));
    if (c.masked) { CK(hipMalloc(&len, c.b * 4)); CK(hipMemcpy(len, hlen.data(), c.b * 4, hipMemcpyHostToDevice)); }
    const int64_t st = c.h * c.d;
    const float scale = 1.f / sqrtf((float)c.d);
    const uint64_t seed = 0x1234567ull;
    // exact-f32 reference (same dropout hash -> same mask)
    GK(gmlm_attention_fwd(qf, kf, vf, len, c.b, c.h, c.l, c.l, c.d, st, st, st, scale, c.drop, seed, nullptr, of, nullptr, lsef, GMLM_F32, nullptr, 0, nullptr, 0, nullptr));
    if (do_bwd) GK(gmlm_attention_bwd(qf, kf, vf, of, gof, lsef, len, c.b, c.h, c.l, c.l, c.d, st, st, st, scale, c.drop, seed, nullptr, dqf, dkf, dvf, st, st, st, GMLM_F32, nullptr, 0, ws, wsb, nullptr, nullptr, nullptr, nullptr, 0, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<float> ro(n), rl(rows * c.h), rdq(n), rdk(n), rdv(n);
    CK(hipMemcpy(ro.data(), of, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rl.data(), lsef, rows * c.h * 4, hipMemcpyDeviceToHost));
    if (do_bwd) { CK(hipMemcpy(rdq.data(), dqf, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rdk.data(), dkf, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rdv.data(), dvf, n * 4, hipMemcpyDeviceToHost)); }
    double useful = 0;
    for (int64_t bb = 0; bb < c.b; ++bb) useful += 4.0 * c.h * c.l * (c.masked ? hlen[bb] : c.l) * c.d;
    const double padded = 4.0 * c.b * c.h * c.l * c.l * c.d;
    for (int var : variants) {
      setenv("GMLM_ATTN_VARIANT", std::to_string(var).c_str(), 1);
      CK(hipMemset(o, 0xFF, n * 2)); CK(hipMemset(lse, 0xFF, rows * c.h * 4));
      GK(gmlm_attention_fwd(q, k, v, len, c.b, c.h, c.l, c.l, c.d, st, st, st, scale, c.drop, seed, nullptr, o, nullptr, lse, GMLM_BF16, nullptr, 0, nullptr, 0, nullptr));
      CK(hipDeviceSynchronize());
      std::vector<uint16_t> ho(n); std::vector<float> hl(rows * c.h);
      CK(hipMemcpy(ho.data(), o, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl.data(), lse, rows * c.h * 4, hipMemcpyDeviceToHost));
      double eo = 0, el = 0; int64_t bad = 0;
      for (int64_t i = 0; i < n; ++i) { const float x = bf2f(ho[i]); if (!(x == x)) ++bad; const double e = fabs((double)x - ro[i]); if (e > eo) eo = e; }
      for (int64_t i = 0; i < rows * c.h; ++i) { const double e = fabs((double)hl[i] - rl[i]); if (!(e == e)) ++bad; else if (e > el) el = e; }
#ifdef GMLM_ATTN_STAMP
      {
        const int SL = 12;
        const size_t nwaves = (size_t)(c.l / 32 + 8) * c.b * c.h;
        uint64_t* dbg; CK(hipMalloc(&dbg, nwaves * SL * 8)); CK(hipMemset(dbg, 0, nwaves * SL * 8));
        gmlm_debug_set_stamp_buffer(dbg);
        GK(gmlm_attention_fwd(q, k, v, len, c.b, c.h, c.l, c.l, c.d, st, st, st, scale, c.drop, seed, nullptr, o, nullptr, lse, GMLM_BF16, nullptr, 0, nullptr, 0, nullptr));
        CK(hipDeviceSynchronize());
        gmlm_debug_set_stamp_buffer(nullptr);
        std::vector<uint64_t> hd(nwaves * SL); CK(hipMemcpy(hd.data(), dbg, nwaves * SL * 8, hipMemcpyDeviceToHost));
        double s[6] = {0, 0, 0, 0, 0, 0}; size_t cnt = 0;
        // s_memtime counters are per XCD: normalise every wave's stamps to the earliest stamp of ITS XCD
        uint64_t xmin[16]; for (auto& x : xmin) x = ~0ull;
        for (size_t wv = 0; wv < nwaves; ++wv) if (hd[wv * SL + 4]) { const int xc = (int)(hd[wv * SL + 11] >> 32) & 15; if (hd[wv * SL + 6] < xmin[xc]) xmin[xc] = hd[wv * SL + 6]; }
        uint64_t tmax = 0;
        double pro = 0, mainl = 0, tail = 0, epi = 0;
        for (size_t wv = 0; wv < nwaves; ++wv) if (hd[wv * SL + 4]) {
          const uint64_t* d = &hd[wv * SL];
          for (int t = 0; t < 6; ++t) s[t] += (double)d[t];
          ++cnt;
          const uint64_t x0 = xmin[(int)(d[11] >> 32) & 15];
          if (d[10] - x0 > tmax) tmax = d[10] - x0;
          pro += (double)(d[7] - d[6]); mainl += (double)(d[8] - d[7]); tail += (double)(d[9] - d[8]); epi += (double)(d[10] - d[9]);
        }
        if (cnt) {
          printf("%-14s var=%d stamps over %zu waves (mean ticks per wave): total %.0f | loop: load-issue %.0f  steps %.0f  dma-wait %.0f  barrier %.0f | blocks/wave %.1f -> ticks per 32-key step %.0f\n",
                 c.tag, var, cnt, s[4] / cnt, s[0] / cnt, s[1] / cnt, s[2] / cnt, s[3] / cnt, s[5] / cnt, s[1] / cnt / (s[5] / cnt));
          printf("%-14s   kernel span %llu ticks (per-XCD clocks aligned at their first wave); per wave: prologue %.0f  main loop %.0f  tail %.0f  epilogue %.0f\n", c.tag,
                 (unsigned long long)tmax, pro / cnt, mainl / cnt, tail / cnt, epi / cnt);
          const int NBIN = 20; double alive[NBIN] = {0}; int starts[NBIN] = {0};
          const double span = (double)tmax;
          for (size_t wv = 0; wv < nwaves; ++wv) if (hd[wv * SL + 4]) {
            const uint64_t* d = &hd[wv * SL];
            const uint64_t x0 = xmin[(int)(d[11] >> 32) & 15];
            const double a = (double)(d[6] - x0) / span * NBIN, b = (double)(d[10] - x0) / span * NBIN;
            int sb = (int)a; if (sb >= NBIN) sb = NBIN - 1; ++starts[sb];
            for (int bn = 0; bn < NBIN; ++bn) { const double lo = bn > a ? bn : a, hi = bn + 1 < b ? bn + 1 : b; if (hi > lo) alive[bn] += hi - lo; }
          }
          printf("%-14s   waves/SIMD alive per 1/20 of the span:", c.tag);
          for (int bn = 0; bn < NBIN; ++bn) printf(" %.1f", alive[bn] / 1024.0);
          printf("\n%-14s   wave starts per bin:", c.tag);
          for (int bn = 0; bn < NBIN; ++bn) printf(" %d", starts[bn]);
          printf("\n");
          // one CU's timeline: the workgroups (wave 0 of each) that ran on the CU of the first stamped wave
          uint32_t hw0 = 0; int xc0 = -1; int shown = 0;
          for (size_t wv = 0; wv < nwaves && shown < 12; wv += 4) if (hd[wv * SL + 4]) {
            const uint64_t* d = &hd[wv * SL];
            const uint32_t hw = (uint32_t)d[11] & 0x0000ff00u /* CU_ID[11:8] + SH/SE bits below are in [15:12] */; const int xc = (int)(d[11] >> 32) & 15;
            const uint32_t cu = ((uint32_t)d[11] >> 8) & 0xf, se = ((uint32_t)d[11] >> 13) & 0x7;
            if (xc0 < 0) { xc0 = xc; hw0 = (se << 4) | cu; }
            if (xc != xc0 || ((se << 4) | cu) != hw0) continue;
            const uint64_t x0 = xmin[xc];
            printf("%-14s   CU(xcc %d se %u cu %u) wg %zu: begin %llu  main %llu  tail %llu  loop-end %llu  end %llu\n", c.tag, xc, se, cu, wv / 4,
                   (unsigned long long)(d[6] - x0), (unsigned long long)(d[7] - x0), (unsigned long long)(d[8] - x0), (unsigned long long)(d[9] - x0), (unsigned long long)(d[10] - x0));
            ++shown;
          }
          (void)hw0;
        }
        (void)hipFree(dbg);
      }
#endif
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = 20;
      for (int i = 0; i < 3; ++i) GK(gmlm_attention_fwd(q, k, v, len, c.b, c.h, c.l, c.l, c.d, st, st, st, scale, c.drop, seed, nullptr, o, nullptr, lse, GMLM_BF16, nullptr, 0, nullptr, 0, nullptr));
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) GK(gmlm_attention_fwd(q, k, v, len, c.b, c.h, c.l, c.l, c.d, st, st, st, scale, c.drop, seed, nullptr, o, nullptr, lse, GMLM_BF16, nullptr, 0, nullptr, 0, nullptr));
      CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
      printf("%-14s var=%d fwd %8.1f us  %7.1f TF/s padded (%.3f of 2.5 PF)  %7.1f TF/s executed   max|o-f32|=%.4f max|lse-f32|=%.5f nan=%ld\n",
             c.tag, var, ms * 1e3, padded / ms / 1e9, padded / ms / 1e9 / 2500.0, useful / ms / 1e9, eo, el, (long)bad);
      if (do_bwd) {
        GK(gmlm_attention_bwd(q, k, v, o, go, lse, len, c.b, c.h, c.l, c.l, c.d, st, st, st, scale, c.drop, seed, nullptr, dq, dk, dv, st, st, st, GMLM_BF16, nullptr, 0, ws, wsb, nullptr, nullptr, nullptr, nullptr, 0, nullptr));
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> g(n);
        double e3[3] = {0, 0, 0}, m3[3] = {0, 0, 0};
        uint16_t* dptr[3] = {dq, dk, dv}; const std::vector<float>* rp[3] = {&rdq, &rdk, &rdv};
        for (int t = 0; t < 3; ++t) {
          CK(hipMemcpy(g.data(), dptr[t], n * 2, hipMemcpyDeviceToHost));
          for (int64_t i = 0; i < n; ++i) { const double x = bf2f(g[i]), e = fabs(x - (*rp[t])[i]); if (!(x == x)) ++bad; if (e > e3[t]) e3[t] = e; if (fabs((*rp[t])[i]) > m3[t]) m3[t] = fabs((*rp[t])[i]); }
        }
        for (int i = 0; i < 3; ++i) GK(gmlm_attention_bwd(q, k, v, o, go, lse, len, c.b, c.h, c.l, c.l, c.d, st, st, st, scale, c.drop, seed, nullptr, dq, dk, dv, st, st, st, GMLM_BF16, nullptr, 0, ws, wsb, nullptr, nullptr, nullptr, nullptr, 0, nullptr));
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) GK(gmlm_attention_bwd(q, k, v, o, go, lse, len, c.b, c.h, c.l, c.l, c.d, st, st, st, scale, c.drop, seed, nullptr, dq, dk, dv, st, st, st, GMLM_BF16, nullptr, 0, ws, wsb, nullptr, nullptr, nullptr, nullptr, 0, nullptr));
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        printf("%-14s var=%d bwd %8.1f us  %7.1f TF/s padded(10/4 x fwd flops) (%.3f)  err dq %.4f/%.2f dk %.4f/%.2f dv %.4f/%.2f nan=%ld\n", c.tag, var, ms * 1e3,
               2.5 * padded / ms / 1e9, 2.5 * padded / ms / 1e9 / 2500.0, e3[0], m3[0], e3[1], m3[1], e3[2], m3[2], (long)bad);
      }
      fflush(stdout);
    }
    (void)hipFree(q); (void)hipFree(k); (void)hipFree(v); (void)hipFree(o); (void)hipFree(go); (void)hipFree(dq); (void)hipFree(dk); (void)hipFree(dv);
    (void)hipFree(qf); (void)hipFree(kf); (void)hipFree(vf); (void)hipFree(of); (void)hipFree(gof); (void)hipFree(dqf); (void)hipFree(dkf); (void)hipFree(dvf);
    (void)hipFree(lse); (void)hipFree(lsef); (void)hipFree(ws); if (len) (void)hipFree(len);
  }
  if (!only || strstr("packed_mix", only[0] == '=' ? only + 1 : only)) {
    // the in-step BERT batch: 1,257 sequences of 16..128 tokens packed back to back (cu_seqlens), fused QKV rows, h = 12, d = 64, p = 0.1
    const int64_t nseq = 1257, h = 12, d = 64, hd = h * d;
    std::vector<int32_t> cu(nseq + 1, 0);
    double pairs = 0;
    std::vector<int> lens(nseq);
    const int lmax_ = getenv("GMLM_BENCH_MAXLEN") ? atoi(getenv("GMLM_BENCH_MAXLEN")) : 128;      // longest sequence of the mix
    const int cls_ = getenv("GMLM_BENCH_CLASSLEN") ? atoi(getenv("GMLM_BENCH_CLASSLEN")) : lmax_;      // max_len handed to the library (capacity class)
    for (auto& x : lens) x = 16 + (int)(rng() % (lmax_ - 15));
    std::sort(lens.begin(), lens.end(), [](int a, int b) { return a > b; });      // the encoder batches by length (descending)
    for (int64_t i = 0; i < nseq; ++i) { cu[i + 1] = cu[i] + lens[i]; pairs += (double)lens[i] * lens[i]; }
    // capacity classes (rows rounded up to 32): contiguous ranges of the sorted batch
    std::vector<std::pair<int64_t, int64_t>> cls;      // (first sequence, count)
    for (int64_t i = 0, j; i < nseq; i = j) { for (j = i; j < nseq && (lens[j] + 31) / 32 == (lens[i] + 31) / 32; ++j) {} cls.push_back({i, j - i}); }
    const int64_t T = cu[nseq];
    std::vector<uint16_t> hqkv(T * 3 * hd), hgo(T * hd);
    for (auto& x : hqkv) x = f2bf(nd(rng) * 0.5f);
    for (auto& x : hgo) x = f2bf(nd(rng));
    uint16_t *qkv, *o, *go, *dqkv; float* lse; int32_t* dcu; void* ws;
    CK(hipMalloc(&qkv, T * 3 * hd * 2)); CK(hipMalloc(&dqkv, T * 3 * hd * 2)); CK(hipMalloc(&o, T * hd * 2)); CK(hipMalloc(&go, T * hd * 2));
    CK(hipMalloc(&lse, T * h * 4)); CK(hipMalloc(&dcu, (nseq + 1) * 4));
    const size_t wsb = gmlm_attention_bwd_workspace_bytes(1, h, T, T, d);
    CK(hipMalloc(&ws, wsb));
    CK(hipMemcpy(qkv, hqkv.data(), T * 3 * hd * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(go, hgo.data(), T * hd * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcu, cu.data(), (nseq + 1) * 4, hipMemcpyHostToDevice));
    const float scale = 0.125f;
    const bool by_class = getenv("GMLM_BENCH_CLASSES") != nullptr;
    // sequence groups (work items of <= 128 rows / <= 13 sequences), as gmlm_amd.ops.pack_sequence_groups builds them
    std::vector<int32_t> grp = {0};
    { int rows = 0, cnt = 0;
      for (int64_t i = 0; i < nseq; ++i) { if (cnt && (rows + lens[i] > 128 || cnt == 13)) { grp.push_back((int32_t)i); rows = cnt = 0; } rows += lens[i]; ++cnt; }
      grp.push_back((int32_t)nseq); }
    const int64_t ngrp = getenv("GMLM_BENCH_NOGROUPS") ? 0 : (int64_t)grp.size() - 1;
    int32_t* dgrp = nullptr;
    if (ngrp) { CK(hipMalloc(&dgrp, grp.size() * 4)); CK(hipMemcpy(dgrp, grp.data(), grp.size() * 4, hipMemcpyHostToDevice)); }
    const float pdrop = getenv("GMLM_BENCH_NODROP") ? 0.f : 0.1f;
    printf("packed_mix     %ld sequences in %ld work items per head%s, dropout %.2f\n", (long)nseq, (long)(ngrp ? ngrp : nseq), ngrp ? " (grouped)" : "", pdrop);
    auto fwd = [&]() {
      if (!by_class) { GK(gmlm_attention_fwd(qkv, qkv + hd, qkv + 2 * hd, nullptr, nseq, h, T, T, d, 3 * hd, 3 * hd, 3 * hd, scale, pdrop, 77, nullptr, o, nullptr, lse, GMLM_BF16, dcu, cls_, dgrp, ngrp, nullptr)); return; }
      for (auto& c : cls)
        GK(gmlm_attention_fwd(qkv, qkv + hd, qkv + 2 * hd, nullptr, c.second, h, T, T, d, 3 * hd, 3 * hd, 3 * hd, scale, pdrop, 77, nullptr, o, nullptr, lse, GMLM_BF16, dcu + c.first, lens[c.first], nullptr, 0, nullptr));
    };
    auto bwd = [&]() {
      if (!by_class) { GK(gmlm_attention_bwd(qkv, qkv + hd, qkv + 2 * hd, o, go, lse, nullptr, nseq, h, T, T, d, 3 * hd, 3 * hd, 3 * hd, scale, pdrop, 77, nullptr,
                                              dqkv, dqkv + hd, dqkv + 2 * hd, 3 * hd, 3 * hd, 3 * hd, GMLM_BF16, dcu, cls_, ws, wsb, nullptr, nullptr, nullptr, dgrp, ngrp, nullptr)); return; }
      for (auto& c : cls)      // one call per capacity class: same tensors, cu_seqlens sub-range, the class's own max_len
        GK(gmlm_attention_bwd(qkv, qkv + hd, qkv + 2 * hd, o, go, lse, nullptr, c.second, h, T, T, d, 3 * hd, 3 * hd, 3 * hd, scale, pdrop, 77, nullptr,
                              dqkv, dqkv + hd, dqkv + 2 * hd, 3 * hd, 3 * hd, 3 * hd, GMLM_BF16, dcu + c.first, lens[c.first], ws, wsb, nullptr, nullptr, nullptr, nullptr, 0, nullptr));
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int i = 0; i < 3; ++i) { fwd(); bwd(); }
    CK(hipEventRecord(e0, nullptr)); for (int i = 0; i < 20; ++i) fwd(); CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); const float f_us = ms / 20 * 1e3f;
    CK(hipEventRecord(e0, nullptr)); for (int i = 0; i < 20; ++i) bwd(); CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); const float b_us = ms / 20 * 1e3f;
    std::vector<uint16_t> hd_(T * 3 * hd); CK(hipMemcpy(hd_.data(), dqkv, T * 3 * hd * 2, hipMemcpyDeviceToHost));
    double cks = 0; long bad = 0; for (auto x : hd_) { const float f = bf2f(x); if (!(f == f)) ++bad; cks += fabs(f); }
    printf("packed_mix     T=%ld: fwd %7.1f us (%.2f TB/s of q,k,v,o)  bwd %7.1f us (%.2f TB/s of 7 tensors: q, k, v, dO in; dq, dk, dv out)  sum|dqkv|=%.6e nan=%ld\n", (long)T, f_us,
           4.0 * T * hd * 2 / f_us / 1e6, b_us, 7.0 * T * hd * 2 / b_us / 1e6, cks, bad);
  }
  return 0;
}
