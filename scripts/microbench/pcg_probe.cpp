// Host channel sampler, piece by piece (build: g++ -O3 -std=c++17 -msse4.1 -ffp-contract=off -pthread pcg_probe.cpp):
// block fill (raw PCG64 outputs + common-path exponentials) over 1..16 threads, the per-channel consumer on the
// prefetched arrays, and the serial blocked stream it replaces - 2.6e6 geometric(0.02) draws each.
#include <cstdarg>
#include <cstdio>
int tsim_fail(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); return code; }
#include "../../tsim_amd/csrc/tsim_pcg.cpp"
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  Pcg g; g.state = 12345; g.inc = 77;
  const int64_t N = 2600000;
  std::vector<uint32_t> frow(3000000);
  for (int T : {1, 2, 4, 8, 16}) {
    Pool pl(T - 1);
    std::vector<uint64_t> raw; std::vector<double> ex;
    double best_fill = 1e9, best_run = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      double t0 = now();
      std::vector<uint8_t> xt; PStream st(g, 2700000, &pl, raw, ex, xt);
      st.close();
      double t1 = now();
      st.ready_upto = st.n_blocks * PStream::kBlk;
      int64_t i = 0, pos = -1; size_t n = 0;
      geometric_run(st, log1p(-0.02), N, 1000000001, 1000000000, frow.data(), frow.size() - 1, i, pos, n);
      double t2 = now();
      if (rep) { best_fill = std::min(best_fill, t1 - t0); best_run = std::min(best_run, t2 - t1); }
    }
    printf("threads %2d: fill of 2.7e6 outputs %.2f ms, consumer (2.6e6 draws, after the fill) %.2f ms\n", T, best_fill, best_run);
  }
  double best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    Stream st(g);
    int64_t i = 0, pos = -1; size_t n = 0;
    double t0 = now();
    geometric_run(st, log1p(-0.02), N, 1000000001, 1000000000, frow.data(), frow.size() - 1, i, pos, n);
    best = std::min(best, now() - t0);
  }
  printf("serial blocked stream: 2.6e6 draws %.2f ms\n", best);
}
