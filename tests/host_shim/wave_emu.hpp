// tests/host_shim/wave_emu.hpp — TEST-ONLY emulation of one 64-lane wavefront with one host thread per lane, so that
// the wave-cooperative device code (nimblephysics_amd/csrc/coop_dev.hpp) can be unit-tested without a GPU.  Cross-lane
// primitives exchange through a shared slot array between two barriers; LDS is plain shared memory.  Control flow in
// the device code is wave-uniform wherever a primitive is called, which is exactly what this emulation requires.
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <math.h>
#include <functional>
#include <thread>
#include <vector>

struct EmuShared {
  pthread_barrier_t bar;
  double dslot[64];
  int bslot[64];
};

struct EmuWave {
  EmuShared* sh;
  int ln;
  int lane() const { return ln; }
  void barrier() const { pthread_barrier_wait(&sh->bar); }
  void sync() const { barrier(); }
  double maxAll(double v) const {
    sh->dslot[ln] = v; barrier();
    double m = sh->dslot[0];
    for (int i = 1; i < 64; i++) m = fmax(m, sh->dslot[i]);
    barrier();
    return m;
  }
  template <int ROWS> double minRows(double v) const { return -maxAll(-v); }
  uint64_t ballot(bool p) const {
    sh->bslot[ln] = p ? 1 : 0; barrier();
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) if (sh->bslot[i]) m |= 1ull << i;
    barrier();
    return m;
  }
  int shflI(int v, int src) const {
    sh->bslot[ln] = v; barrier();
    const int r = sh->bslot[src & 63];
    barrier();
    return r;
  }
  double bcast(double v, int src) const { return shfl(v, src); }
  int bcastI(int v, int src) const { return shflI(v, src); }
  double shfl(double v, int src) const {
    sh->dslot[ln] = v; barrier();
    const double r = sh->dslot[src & 63];
    barrier();
    return r;
  }
};

// run body(wave) on 64 lane-threads
inline void emuRunWave(const std::function<void(const EmuWave&)>& body) {
  EmuShared sh;
  pthread_barrier_init(&sh.bar, nullptr, 64);
  std::vector<std::thread> th;
  for (int l = 0; l < 64; l++) th.emplace_back([&sh, l, &body]() { EmuWave w{&sh, l}; body(w); });
  for (auto& t : th) t.join();
  pthread_barrier_destroy(&sh.bar);
}
