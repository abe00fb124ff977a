// k22 — on-device selection of conv / GEMM tile configurations, shared by the UNet, MoVQ and prior engines.
//
// A Tuned describes one launch_igemm problem plus the list of configurations worth trying; tune_igemm_ops() times
// every candidate of every DISTINCT problem on the device (with the Infinity Cache flushed between runs: in a real
// step the weights stream from HBM) and keeps the fastest.  Optional persistent cache: env K22_TUNE_CACHE=<file>.
#pragma once
#include "kernels.h"

#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>
#include <stdio.h>
#include <stdlib.h>

// One tile configuration of launch_igemm: algo 1 = generic implicit GEMM (bm x bn tile, LDS-DMA depth `stages`),
// algo 2 = LDS-resident halo kernel for 3x3 convolutions (bm = 256 / 128); splitk >= 1.
struct Cfg { int algo = 0, bm = 0, bn = 0, splitk = 0, stages = 0; };

struct TunedSlot;  // engine-specific workspace slot (opaque here)

// ---- the tile table --------------------------------------------------------------------------------------------------
// One process-wide map  problem -> tile configuration, shared by the UNet, MoVQ and prior engines.  It is filled from
//   (1) the table SHIPPED with the package (kandinsky-2_amd/tiles_gfx950.txt, loaded by k22_tile_table_load when the
//       library is opened): the configurations measured once on an MI355X for the shapes of BASELINE.json's configs.  A
//       problem found here is never timed again, so the same build gives the same bits on every box (split-K factors and
//       GroupNorm partial-sum row counts, i.e. the fp32 summation orders, are part of the configuration);
//   (2) env K22_TUNE_CACHE=<file> (merged on first use, rewritten with every dtype's lines when something new was measured);
//   (3) on-device measurement of problems that are in neither (autotune on, the default).  K22_AUTOTUNE=0 replaces (3)
//       by the fixed heuristic: strict run-to-run and box-to-box reproducibility for shapes outside the shipped table.
// Key: dtype, taps, M, N, Kc, K0 (+ fused-skip width), H, W, output mode (+ residual / activation flags), stats wanted.
typedef std::tuple<int, int, int, int, int, int, int, int, int, bool> TileKey;
struct TileTable {
  std::map<TileKey, std::pair<Cfg, float>> m;   // value: configuration, measured time in ms (0 = unknown)
  std::mutex mu;
  bool env_loaded = false;
  size_t measured = 0;                           // entries added by measurement in this process
};
inline TileTable& tile_table() { static TileTable t; return t; }

// merges the lines of `path` into the table (later lines win); returns the number of lines read, -1 if unreadable
inline int tile_table_load_file(const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  TileTable& tt = tile_table();
  std::lock_guard<std::mutex> lk(tt.mu);
  int n = 0;
  char line[512];
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '#' || line[0] == '\n') continue;
    int dtp, taps, M, N, Kc, K0, H, W, om, ws, algo, bm, bn, sk, stg; float us;
    if (sscanf(line, "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %f", &dtp, &taps, &M, &N, &Kc, &K0, &H, &W, &om, &ws,
               &algo, &bm, &bn, &sk, &stg, &us) != 16) continue;
    Cfg c; c.algo = algo; c.bm = bm; c.bn = bn; c.splitk = sk; c.stages = stg;
    tt.m[TileKey(dtp, taps, M, N, Kc, K0, H, W, om, ws != 0)] = std::make_pair(c, us * 1e-3f);
    ++n;
  }
  fclose(f);
  return n;
}
inline int tile_table_save_file(const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  TileTable& tt = tile_table();
  std::lock_guard<std::mutex> lk(tt.mu);
  fprintf(f, "# k22 tile table: dtype taps M N Kc K0 H W outmode stats | algo bm bn splitk stages | time_us\n");
  for (auto& kv : tt.m) {
    const TileKey& k = kv.first; const Cfg& c = kv.second.first;
    fprintf(f, "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %.2f\n", std::get<0>(k), std::get<1>(k), std::get<2>(k), std::get<3>(k),
            std::get<4>(k), std::get<5>(k), std::get<6>(k), std::get<7>(k), std::get<8>(k), std::get<9>(k) ? 1 : 0,
            c.algo, c.bm, c.bn, c.splitk, c.stages, kv.second.second * 1e3f);
  }
  fclose(f);
  return (int)tt.m.size();
}
inline void tile_table_load_env_once() {
  TileTable& tt = tile_table();
  bool need;
  { std::lock_guard<std::mutex> lk(tt.mu); need = !tt.env_loaded; tt.env_loaded = true; }
  if (need) if (const char* cp = getenv("K22_TUNE_CACHE")) (void)tile_table_load_file(cp);
}

// A conv / GEMM launch whose configuration is chosen by measurement on the device (first forward of a plan).
struct Tuned {
  IgemmParams p = {};          // problem; device pointers are filled in at launch time
  std::vector<Cfg> cands;
  Cfg cfg;                     // current choice (heuristic until tuned)
  bool want_stats = false;     // epilogue also emits the GroupNorm partial sums of its output
  int rpi = 0;                 // stats rows per image under cfg
  float best_us = 0.f;
  bool from_table = false;     // cfg came from the tile table (shipped / cache / measured earlier in this process)
  int dt = -1;                 // arithmetic of THIS op when it differs from its engine's (a K22_F16X2 plan mixes x2 and x3 ops); -1 = the engine's
  void* aux0 = nullptr; void* aux1 = nullptr;  // engine-specific (IG_OUT_QKV: this block's K_all / V^T_all slots)
  std::function<int(hipStream_t)> run;
};

inline void tuned_apply_cfg(IgemmParams& q, const Cfg& c) {
  q.algo = c.algo; q.force_bm = c.bm; q.force_bn = c.bn; q.splitk = c.splitk; q.stages = c.stages;
}

inline void tuned_make_candidates(Tuned& t, int dtype) {
  const IgemmParams& p = t.p;
  const int BK = k22_bk(dtype);
  const int nkt = p.taps * (p.Kc / BK);
  std::vector<Cfg> all;
  if (p.taps == 9) {
    const int B = p.M / (p.H * p.W);
    // algo 2 = lock-step halo kernel; 3 = 64-byte rows / filter-row iterations (only ever wins at the smallest level,
    // deep split-K); 4 = two-phase kernel (measured 12-20 % slower than 2 on every shape: not a candidate)
    // 5 = loader-wave specialisation (measured 2-5 % slower: not a candidate); 7 = LDS-DMA from inline asm (exact
    // lgkmcnt for the fragment reads, +1-3 %); 6 = 7 + explicit fragment pipeline across the barrier
    // 11 = producer / consumer wave specialisation (conv3_halo_spec_kernel)
    for (int algo : {2, 7, 6, 3, 11, 12}) {
      if (algo == 3 && p.H > 16) continue;
      if (k22_is_split(dtype) && (algo == 2 || algo == 6)) continue;   // split precision: one lock-step form (7) + the specialised ones
      IgemmParams ph = p;
      ph.algo = algo;
      const int nsplit_max = (p.Kc / BK) * (algo == 3 ? 2 : 1);
      for (int bm : {256, 128}) {
        if (!conv3_halo_supported(ph, dtype, bm) || p.N < 128) continue;
        const int nb = B * conv3_halo_tiles_per_image(p, bm) * ((p.N + 127) / 128);
        for (int sk : {1, 2, 3, 4, 5, 6, 8, 10, 12}) {
          if (sk > nsplit_max || (sk > 1 && nb * sk > 800) || (sk > 1 && nb >= 256)) continue;
          Cfg c; c.algo = algo; c.bm = bm; c.bn = 0; c.splitk = sk; c.stages = 0;
          all.push_back(c);
        }
      }
    }
  }
  if (p.taps == 1) {
    // 8-wave BM x 128 tile kernel (gemm8_kernel)
    const int hw = p.H > 0 ? p.H * p.W : p.M;
    for (int bm : {256, 128}) {
      IgemmParams q = p;
      q.algo = 10; q.force_bm = bm;
      if (!gemm8_supported(q, dtype, bm)) continue;
      const int nb = (p.M / hw) * gemm8_tiles_per_image(p, bm) * ((p.N + 127) / 128);
      for (int sk : {1, 2, 3, 4, 6}) {
        if (sk > 1 && (nkt / sk < 4 || nb * sk > 800 || nb >= 256 || p.out_mode == IG_OUT_QKV)) continue;
        for (int stg : {0, 2, 3, 4}) {   // 2: two co-resident workgroups per CU with a 2-deep ring (BM = 128 only); 3 / 4: gemm8_spec_kernel (16-bit types), 4 = its two-per-CU form
          if ((stg == 2 || stg == 4) && bm != 128) continue;
          if (stg >= 3 && !gemm8_spec_supported(dtype)) continue;
          Cfg c; c.algo = 10; c.bm = bm; c.bn = 0; c.splitk = sk; c.stages = stg;
          all.push_back(c);
        }
      }
    }
  }
  // weight-streaming small-M kernel (stream_gemm.hip): 5 / 9 m-blocks per workgroup, 64-wide n-tiles, any split-K.  A candidate only
  // when the owner of the weights made the fragment-major copy (the row-major form of the kernel never won a measurement) and for the 3x3
  // convolutions (at taps == 1 gemm8 / igemm were faster on every GEMM of the UNet, profiles/r03_stream_kernel_v2.txt)
  if (k22_esz(dtype) == 2 && p.Wfrag != nullptr && p.taps == 9) {
    for (int mb : {5, 9}) {
      IgemmParams q = p;
      q.algo = 20; q.force_bm = mb * 32;
      if (!stream_supported(q, dtype, mb)) continue;
      const int nb = stream_mtiles(q, mb);
      if (nb > 320) continue;
      const int nslab = p.Kc / 64;
      for (int sk : {1, 2, 3, 4, 5, 6, 8, 10, 12, 16}) {
        if (sk > nslab || (sk > 1 && nb * sk > 320)) continue;
        Cfg c; c.algo = 20; c.bm = mb * 32; c.bn = 0; c.splitk = sk; c.stages = 0;
        all.push_back(c);
      }
    }
  }
  const int tiles[3][2] = {{128, 128}, {128, 64}, {64, 64}};
  // GEMMs whose weights stream from HBM are latency bound per workgroup (one 8-16 KB tile per round trip with a
  // 2-deep ring): also try deeper rings
  const bool skinny = p.taps == 1;
  for (auto& tl : tiles) {
    if (p.S0 != nullptr) break;  // a fused skip connection rides on the halo kernel only
    if (tl[1] == 128 && p.N <= 64) continue;
    if (tl[0] == 128 && p.M <= 64) continue;
    const int nb = ((p.M + tl[0] - 1) / tl[0]) * ((p.N + tl[1] - 1) / tl[1]);
    for (int sk : {1, 2, 4, 8, 16}) {
      if (sk > 1 && (nkt / sk < 4 || nb * sk > 1536 || nb >= 384 || p.out_mode == IG_OUT_QKV)) continue;
      for (int stg : {2, 3, 4}) {
        if (stg > 2 && !skinny) continue;
        if (stg * (tl[0] + tl[1]) * 128 > 160 * 1024) continue;
        Cfg c; c.algo = 1; c.bm = tl[0]; c.bn = tl[1]; c.splitk = sk; c.stages = stg;
        all.push_back(c);
      }
    }
  }
  // keep what can deliver the requested side output and whose split-K scratch stays reasonable
  t.cands.clear();
  for (auto& c : all) {
    IgemmParams q = p;
    tuned_apply_cfg(q, c);
    if ((size_t)c.splitk * p.M * p.N * sizeof(float) > ((size_t)96 << 20) && c.splitk > 1) continue;
    if (t.want_stats && igemm_stats_rows_per_image(q, dtype) <= 0) continue;
    t.cands.push_back(c);
  }
}

inline TileKey tuned_key(const Tuned& t, int dtype) {
  const IgemmParams& p = t.p;
  // fp16 runs the same kernels on the same bytes at the same MFMA rate as bf16: one table line serves both 16-bit types
  if (dtype == K22_F16) dtype = K22_BF16;
  // the asymmetric split runs the split-precision kernels' frames on the same bytes with two of their three MFMAs: it resolves through
  // their lines unless a line of its own exists (tile_table_lookup tries its own key first)
  return TileKey(dtype, p.taps, p.M, p.N, p.Kc, p.K0 + (p.S0 ? 100000 * (p.SK0 + p.SK1) : 0), p.H, p.W,
                 p.out_mode + 16 * p.res_f32 + 32 * p.act + 256 * p.a_raw, t.want_stats);
}

// a table line from an older build may name a configuration this build would not generate: only candidates are accepted
inline bool tuned_is_candidate(const Tuned& t, const Cfg& c) {
  for (auto& k : t.cands)
    if (k.algo == c.algo && k.bm == c.bm && k.bn == c.bn && k.splitk == c.splitk && (k.stages == c.stages || (c.algo >= 2 && c.algo != 10))) return true;
  return false;
}

inline bool tile_table_lookup(const Tuned& t, int dtype, Cfg* out, float* us) {
  tile_table_load_env_once();
  TileTable& tt = tile_table();
  std::lock_guard<std::mutex> lk(tt.mu);
  auto it = tt.m.find(tuned_key(t, dtype));
  // K22_X2_OWN_LINES=1 (tools/make_tile_table.py --x2-only): no fall-back, so that the asymmetric split's own lines get measured
  static const bool x2_own = getenv("K22_X2_OWN_LINES") && atoi(getenv("K22_X2_OWN_LINES")) != 0;
  if ((it == tt.m.end() || !tuned_is_candidate(t, it->second.first)) && dtype == K22_F16X2 && !x2_own) it = tt.m.find(tuned_key(t, K22_F16X3));
  if (it == tt.m.end() || !tuned_is_candidate(t, it->second.first)) return false;
  *out = it->second.first;
  if (us) *us = it->second.second * 1e3f;
  return true;
}

inline void tuned_finish_cfg(Tuned& t, int dtype) {
  IgemmParams q = t.p;
  tuned_apply_cfg(q, t.cfg);
  t.rpi = t.want_stats ? igemm_stats_rows_per_image(q, dtype) : 0;
}

// heuristic starting point (also the final choice when auto-tuning is off)
inline void tuned_default_cfg(Tuned& t, int dtype) {
  IgemmParams q = t.p;
  q.algo = 0; q.force_bm = 0; q.force_bn = 0; q.splitk = 0; q.stages = 0;
  Cfg c; c.algo = 0; c.bm = 0; c.bn = 0; c.splitk = igemm_choose_splitk(q, dtype); c.stages = 0;
  if (q.out_mode == IG_OUT_QKV) c.splitk = 1;
  tuned_apply_cfg(q, c);
  if (t.want_stats && igemm_stats_rows_per_image(q, dtype) <= 0) {
    if (t.cands.empty()) { t.want_stats = false; }
    else c = t.cands[0];
  }
  t.cfg = c;
  t.from_table = false;
  if (t.p.M >= 64 && t.cands.size() >= 2) {
    Cfg tc; float us = 0.f;
    if (tile_table_lookup(t, dtype, &tc, &us)) { t.cfg = tc; t.best_us = us; t.from_table = true; }
  }
  tuned_finish_cfg(t, dtype);
}

inline size_t tuned_max_splitk_bytes(const Tuned& t, bool autotune) {
  size_t m = 0;
  auto upd = [&](const Cfg& c) { if (c.splitk > 1 || c.algo == 20) m = std::max(m, (size_t)c.splitk * t.p.M * t.p.N * sizeof(float)); };
  upd(t.cfg);
  if (autotune) for (auto& c : t.cands) upd(c);
  return m;
}

inline int tuned_max_rpi(const Tuned& t, int dtype, bool autotune) {
  int m = t.rpi;
  if (autotune) for (auto& c : t.cands) { IgemmParams q = t.p; tuned_apply_cfg(q, c); m = std::max(m, igemm_stats_rows_per_image(q, dtype)); }
  return m;
}

// Measures every candidate of every distinct problem that the tile table does not know yet (outputs written meanwhile are
// garbage; the caller runs the real forward afterwards).  flush / flush_bytes: a scratch region memset between runs to
// evict the Infinity Cache.  Problems already resolved from the table (Tuned::from_table) are left alone.
inline int tune_igemm_ops(std::deque<Tuned>& tuned, int engine_dtype, void* flush, size_t flush_bytes, hipStream_t st) {
  tile_table_load_env_once();
  TileTable& tt = tile_table();
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return k22_set_error(K22_EHIP, "tune: hipEventCreate");
  int rc = K22_OK;
  size_t added = 0;
  for (auto& t : tuned) {
    if (!t.run || t.cands.size() < 2) continue;
    const IgemmParams& p = t.p;
    if (p.M < 64) continue;
    const int dtype = t.dt >= 0 ? t.dt : engine_dtype;
    Cfg hit; float hit_us = 0.f;
    if (tile_table_lookup(t, dtype, &hit, &hit_us)) {
      t.cfg = hit; t.best_us = hit_us; t.from_table = true;
      tuned_finish_cfg(t, dtype);
      continue;
    }
    Cfg best = t.cfg; float best_ms = 1e30f;
    for (auto& c : t.cands) {
      t.cfg = c;
      tuned_finish_cfg(t, dtype);
      float tmin = 1e30f;
      // rep 0 = warm-up (code load, function attributes); the minimum of the next K22_TUNE_REPS (default 5) is kept: a whole
      // table costs about a second, and a noisy pick stays for the life of the process
      static const int timed_reps = [] { const char* e = getenv("K22_TUNE_REPS"); const int v = e ? atoi(e) : 5; return v < 1 ? 1 : (v > 20 ? 20 : v); }();
      for (int rep = 0; rep <= timed_reps && rc == K22_OK; ++rep) {
        if (flush && flush_bytes) (void)hipMemsetAsync(flush, 0, flush_bytes, st);
        (void)hipEventRecord(e0, st);
        rc = t.run(st);
        (void)hipEventRecord(e1, st);
        if (hipStreamSynchronize(st) != hipSuccess) rc = k22_set_error(K22_EHIP, "tune: kernel failed");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < tmin) tmin = ms;
      }
      if (rc) break;
      if (tmin < best_ms) { best_ms = tmin; best = c; }
    }
    if (rc) break;
    t.cfg = best; t.best_us = best_ms * 1e3f; t.from_table = true;
    tuned_finish_cfg(t, dtype);
    {
      std::lock_guard<std::mutex> lk(tt.mu);
      tt.m[tuned_key(t, dtype)] = std::make_pair(best, best_ms);
      tt.measured++;
    }
    ++added;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc == K22_OK && added) if (const char* cp = getenv("K22_TUNE_CACHE")) (void)tile_table_save_file(cp);
  return rc;
}

inline std::string tuning_report_text(const std::deque<Tuned>& tuned) {
  std::string out = "taps      M     N     K    H    W stats | algo  bm  bn splitk stg |  time_us  count\n";
  std::map<std::string, int> seen;
  std::vector<std::string> order;
  for (auto& t : tuned) {
    char line[256];
    snprintf(line, sizeof line, "%4d %6d %5d %5d %4d %4d %5d | %4s %3d %3d %6d %3d | %8.1f", t.p.taps, t.p.M, t.p.N, t.p.Kc,
             t.p.H, t.p.W, t.want_stats ? 1 : 0, t.cfg.algo == 2 ? "halo" : (t.cfg.algo == 3 ? "hal3" : (t.cfg.algo == 4 ? "hal4" : (t.cfg.algo == 5 ? "hal5" : (t.cfg.algo == 6 ? "hal6" : (t.cfg.algo == 7 ? "hal7" : (t.cfg.algo == 10 ? "gem8" : (t.cfg.algo == 11 ? "spec" : (t.cfg.algo == 12 ? "spcp" : (t.cfg.algo == 1 ? "gen" : "auto"))))))))), t.cfg.bm, t.cfg.bn,
             t.cfg.splitk, t.cfg.stages, t.best_us);
    if (!seen.count(line)) order.push_back(line);
    seen[line]++;
  }
  for (auto& l : order) { out += l; out += "  x" + std::to_string(seen[l]) + "\n"; }
  return out;
}
