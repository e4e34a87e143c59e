// power_trace.c — sample the GPU's clocks and socket power as fast as the SMI metrics table updates, while another
// process runs a workload.  Evidence for / against the "the step runs clocked down at the power limit" reading of the
// flat C2b step: prints one CSV row per NEW firmware sample.
//
//   gcc -O2 tools/power_trace.c -I/opt/rocm/include -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib -o /tmp/power_trace
//   /tmp/power_trace <seconds> [device] > trace.csv &      # then start the workload
//
// Columns: t_ms (host monotonic), fw_ts (firmware timestamp, 10 ns units), gfxclk_mhz (mean of the 8 XCD clocks), gfxclk_min,
// gfxclk_max, uclk_mhz, power_w (current socket power), gfx_busy (average_gfx_activity, %), umc_busy (%), ppt_acc (accumulated
// power-throttle residency), thm_acc, energy (accumulator; 15.3 uJ units), temp_hot.
#include <rocm_smi/rocm_smi.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 10.0;
  const unsigned dev = argc > 2 ? (unsigned)atoi(argv[2]) : 0u;
  if (rsmi_init(0) != RSMI_STATUS_SUCCESS) { fprintf(stderr, "rsmi_init failed\n"); return 1; }
  printf("t_ms,fw_ts,gfxclk_mhz,gfxclk_min,gfxclk_max,uclk_mhz,power_w,gfx_busy,umc_busy,ppt_acc,thm_acc,energy,temp_hot\n");
  const double t0 = now_ms();
  unsigned long long last_ts = 0;
  long reads = 0, rows = 0;
  while (now_ms() - t0 < secs * 1e3) {
    rsmi_gpu_metrics_t m;
    if (rsmi_dev_gpu_metrics_info_get(dev, &m) != RSMI_STATUS_SUCCESS) { fprintf(stderr, "metrics read failed\n"); break; }
    ++reads;
    if (m.firmware_timestamp == last_ts) continue;
    last_ts = m.firmware_timestamp;
    unsigned lo = 65535, hi = 0, n = 0;
    double sum = 0;
    for (int i = 0; i < RSMI_MAX_NUM_GFX_CLKS; ++i) {
      const unsigned c = m.current_gfxclks[i];
      if (c == 0 || c == 65535) continue;
      sum += c; ++n;
      if (c < lo) lo = c;
      if (c > hi) hi = c;
    }
    printf("%.3f,%llu,%.0f,%u,%u,%u,%u,%u,%u,%llu,%llu,%llu,%u\n", now_ms() - t0, (unsigned long long)m.firmware_timestamp,
           n ? sum / n : (double)m.current_gfxclk, n ? lo : 0, n ? hi : 0, (unsigned)m.current_uclk, (unsigned)m.current_socket_power,
           (unsigned)m.average_gfx_activity, (unsigned)m.average_umc_activity, (unsigned long long)m.ppt_residency_acc,
           (unsigned long long)m.socket_thm_residency_acc, (unsigned long long)m.energy_accumulator, (unsigned)m.temperature_hotspot);
    ++rows;
  }
  fprintf(stderr, "power_trace: %ld reads, %ld distinct firmware samples in %.1f s\n", reads, rows, (now_ms() - t0) * 1e-3);
  rsmi_shut_down();
  return 0;
}
