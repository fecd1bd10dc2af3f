// Which CUs does a CU-masked stream reach?  (evidence for the partitioning in blr_abi.hip pipeline_streams)
//   hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip && ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
__global__ void where(uint32_t* out) {
  extern __shared__ char dyn[];
  if (threadIdx.x == 9999) dyn[0] = 1;
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the workgroup alive for a while so that the launch spreads over every CU the queue may use
  uint64_t t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 200000) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
static void run(const char* name, hipStream_t st) {
  const int n = 4096;
  uint32_t* d; hipMalloc(&d, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, st, d);
  hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  hipLaunchKernelGGL(where, dim3(n), dim3(256), 65536, st, d);   // 64 KB of LDS: two workgroups per CU, as the Gram kernel
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint32_t> h(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<uint32_t, std::set<uint32_t>> per_xcc;
  for (int i = 0; i < n; ++i) {
    uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  int total = 0;
  printf("%-28s", name);
  for (auto& kv : per_xcc) { printf(" x%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
  std::map<uint32_t, int> se_cnt;
  for (uint32_t v : per_xcc[0]) se_cnt[v >> 4]++;
  printf("  total %d | xcc0 CUs per (se,sh):", total);
  for (auto& kv : se_cnt) printf(" %x:%d", kv.first, kv.second);
  printf(" | %.2f ms\n", ms);
  hipFree(d);
}
int main() {
  hipStream_t s;
  hipStreamCreate(&s); run("plain", s);
  auto masked = [&](const char* name, auto pred) {
    uint32_t m[8] = {0};
    for (int i = 0; i < 256; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32);
    hipStream_t ms;
    if (hipExtStreamCreateWithCUMask(&ms, 8, m) != hipSuccess) { printf("%s: create failed\n", name); return; }
    run(name, ms);
    hipStreamDestroy(ms);
  };
  masked("chain {(i/8)%8==i%8}", [](int i) { return ((i / 8) % 8) == (i % 8); });
  masked("gram  complement", [](int i) { return ((i / 8) % 8) != (i % 8); });
  masked("bits 0..127", [](int i) { return i < 128; });
  masked("bits 0..223", [](int i) { return i < 224; });
  masked("bits 32..255", [](int i) { return i >= 32; });
  masked("even bits", [](int i) { return i % 2 == 0; });
  masked("i%8 != 7", [](int i) { return i % 8 != 7; });
  masked("(i/8)%4 != 3", [](int i) { return (i / 8) % 4 != 3; });
  masked("all 256", [](int i) { return true; });
  masked("bits 0..191", [](int i) { return i < 192; });
  masked("q=i/8: q%8!=7", [](int i) { return (i / 8) % 8 != 7; });
  masked("q=i/8: q>=4", [](int i) { return (i / 8) >= 4; });
  masked("q=i/8: q<4", [](int i) { return (i / 8) < 4; });
  masked("q=i/8: q>=28", [](int i) { return (i / 8) >= 28; });
  return 0;
}
