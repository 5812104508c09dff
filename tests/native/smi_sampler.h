// Host-side sampler of socket power and shader clock (librocm_smi64) for the power traces of tests/native (test / bench
// infrastructure).  One sample every ~2 ms from a background thread.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include <rocm_smi/rocm_smi.h>

struct SmiSampler {
  struct S { double t_ms, watts, sclk_mhz; };
  std::vector<S> samples;
  std::atomic<bool> stop{false};
  std::thread th;
  std::chrono::steady_clock::time_point t0;
  bool ok = false;
  void start() {
    ok = rsmi_init(0) == RSMI_STATUS_SUCCESS;
    t0 = std::chrono::steady_clock::now();
    if (!ok) { printf("POWER rocm_smi unavailable: no power trace\n"); return; }
    th = std::thread([this] {
      while (!stop.load()) {
        uint64_t uw = 0; RSMI_POWER_TYPE pt;
        double w = -1, mhz = -1;
        if (rsmi_dev_power_get(0, &uw, &pt) == RSMI_STATUS_SUCCESS) w = uw * 1e-6;
        rsmi_frequencies_t f;
        if (rsmi_dev_gpu_clk_freq_get(0, RSMI_CLK_TYPE_SYS, &f) == RSMI_STATUS_SUCCESS && f.current < RSMI_MAX_NUM_FREQUENCIES) mhz = f.frequency[f.current] * 1e-6;
        samples.push_back({now_ms(), w, mhz});
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    });
  }
  double now_ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
  void finish() { stop.store(true); if (th.joinable()) th.join(); if (ok) rsmi_shut_down(); }
  // mean over [a, b] ms
  void mean(double a, double b, double& w, double& mhz, int& n) const {
    w = mhz = 0; n = 0;
    for (auto& s : samples) if (s.t_ms >= a && s.t_ms <= b && s.watts > 0) { w += s.watts; mhz += s.sclk_mhz; ++n; }
    if (n) { w /= n; mhz /= n; }
  }
  void dump(const char* path, const std::vector<std::pair<double, std::string>>& marks) const {
    FILE* f = fopen(path, "w");
    if (!f) return;
    fprintf(f, "# t_ms,socket_power_W,sclk_MHz   (phase marks below as '# mark t_ms name')\n");
    for (auto& m : marks) fprintf(f, "# mark %.1f %s\n", m.first, m.second.c_str());
    for (auto& s : samples) fprintf(f, "%.2f,%.1f,%.0f\n", s.t_ms, s.watts, s.sclk_mhz);
    fclose(f);
  }
};

