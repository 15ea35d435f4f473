// Resident workgroups per CU against the LDS a workgroup asks for, measured (not computed):
// every workgroup bumps its CU's counter, spins 30 us, records the peak.  This is how the 1,280 B
// allocation unit of gfx950's LDS was found (4 workgroups up to 40,960 B static + dynamic, 3 up to
// 53,760 B; hipOccupancyMaxActiveBlocksPerMultiprocessor divides by the raw size and is one too
// high just below those steps).  The engine sizes its neighbour-list rows by it (cs_engine.hip.inc).
//   hipcc --offload-arch=gfx950 -O2 -o lds_probe tools/lds_probe.hip && ./lds_probe     (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
extern __shared__ unsigned char smem[];
// resident workgroups per CU, measured: every workgroup bumps its CU's counter, waits, records the peak
__global__ void __launch_bounds__(256, 4) k(unsigned* cur, unsigned* peak, float* out) {
  __shared__ unsigned s[72];
  s[threadIdx.x % 72] = threadIdx.x;
  __syncthreads();
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_ID
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));   // XCC_ID[3:0]
  unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
  unsigned key = ((xcc & 0xF) << 8) | (se << 5) | (sh << 4) | cu;
  if (threadIdx.x == 0) {
    unsigned now = atomicAdd(&cur[key], 1u) + 1u;
    atomicMax(&peak[key], now);
  }
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 3000) { }  // 30 us at 100 MHz
  __syncthreads();
  if (threadIdx.x == 0) atomicSub(&cur[key], 1u);
  out[blockIdx.x * 256 + threadIdx.x] = smem[threadIdx.x] + s[(threadIdx.x + 1) % 72];
}
int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  unsigned *cur, *peak; float* out;
  hipMalloc(&cur, 4096 * 4); hipMalloc(&peak, 4096 * 4); hipMalloc(&out, 8192 * 256 * 4);
  for (size_t dyn : {30000, 32000, 32480, 32768, 35000, 40000, 40448, 40672, 40960, 45000, 46000, 48000, 50176, 50800, 51200, 52224, 53248, 54272, 58368, 70000, 80000, 81000}) {
    hipMemset(cur, 0, 4096 * 4); hipMemset(peak, 0, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(8192), dim3(256), dyn, 0, cur, peak, out);
    hipDeviceSynchronize();
    std::vector<unsigned> h(4096); hipMemcpy(h.data(), peak, 4096 * 4, hipMemcpyDeviceToHost);
    unsigned mx = 0, n = 0; unsigned long sum = 0;
    for (unsigned v : h) if (v) { mx = std::max(mx, v); ++n; sum += v; }
    printf("dyn %6zu (+288 static): CUs seen %u, peak resident WGs per CU max %u mean %.2f\n", dyn, n, mx, n ? (double)sum / n : 0.0);
  }
  return 0;
}
