// Probe: how hipExtStreamCreateWithCUMask bits map to XCDs on MI355X, and where the blocks of a grid land.
// Build: hipcc -O2 --offload-arch=gfx950 tools/probes/probe_cumask.hip -o tools/probes/probe_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void fat_kernel(long spin) {
  extern __shared__ double sm[];
  sm[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < spin) __builtin_amdgcn_s_sleep(8);
}

__global__ void where_kernel(int* xcc, int* cu, long spin) {
  if (threadIdx.x == 0) {
    xcc[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((4 - 1) << 11 | (0 << 6) | 20);      // HW_REG_XCC_ID[3:0]
    cu[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((32 - 1) << 11 | (0 << 6) | 4);        // HW_REG_HW_ID
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < spin) __builtin_amdgcn_s_sleep(8);
}

static int run(const char* name, hipStream_t st, int blocks, int threads, long spin) {
  int *dx, *dc;
  CK(hipMalloc(&dx, sizeof(int) * blocks));
  CK(hipMalloc(&dc, sizeof(int) * blocks));
  hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(threads), 0, st, dx, dc, spin);
  CK(hipStreamSynchronize(st));
  std::vector<int> hx(blocks), hc(blocks);
  CK(hipMemcpy(hx.data(), dx, sizeof(int) * blocks, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hc.data(), dc, sizeof(int) * blocks, hipMemcpyDeviceToHost));
  int hist[16] = {0};
  for (int b = 0; b < blocks; ++b) hist[hx[b] & 15]++;
  // distinct CUs seen: HW_ID cu_id[11:8] sh_id[12] se_id[15:13]
  int per_xcc[8] = {0};
  {
    std::vector<int> seen;
    for (int b = 0; b < blocks; ++b) {
      const int key = ((hx[b] & 15) << 16) | ((hc[b] >> 8) & 0xff);
      bool f = false;
      for (int k : seen) f |= (k == key);
      if (!f) { seen.push_back(key); per_xcc[hx[b] & 7]++; }
    }
    printf("  distinct CUs %zu, per xcc:", seen.size());
    for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]);
    printf("\n");
  }
  printf("%-28s blocks=%d xcc histogram:", name, blocks);
  for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
  printf("   first 16 blocks -> xcc:");
  for (int b = 0; b < 16 && b < blocks; ++b) printf(" %d", hx[b]);
  printf("\n");
  hipFree(dx); hipFree(dc);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  hipStream_t plain;
  CK(hipStreamCreate(&plain));
  run("plain stream", plain, 4096, 64, 2000);
  run("plain stream 64x1024", plain, 64, 1024, 20000);
  const int words = 8;  // 256 bits
  struct { const char* name; std::vector<uint32_t> mask; } cases[4];
  cases[0].name = "bits i%8==0 (32 CUs)"; cases[0].mask.assign(words, 0x01010101u);
  cases[1].name = "bits 0..31"; cases[1].mask.assign(words, 0u); cases[1].mask[0] = 0xffffffffu;
  cases[2].name = "bits i%8!=0 (224 CUs)"; cases[2].mask.assign(words, 0xfefefefeu);
  cases[3].name = "bits 32..255"; cases[3].mask.assign(words, 0xffffffffu); cases[3].mask[0] = 0u;
  for (auto& c : cases) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, words, c.mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask -> %s\n", c.name, hipGetErrorString(e)); continue; }
    run(c.name, st, 4096, 64, 2000);
    run((std::string(c.name) + " 1024thr").c_str(), st, 64, 1024, 20000);
    hipStreamDestroy(st);
  }
  // concurrency: long spin on the 32-CU stream, short kernels on the 224-CU stream; do they overlap?
  hipStream_t s0, s1;
  if (hipExtStreamCreateWithCUMask(&s0, words, cases[0].mask.data()) == hipSuccess &&
      hipExtStreamCreateWithCUMask(&s1, words, cases[2].mask.data()) == hipSuccess) {
    int *dx, *dc;
    CK(hipMalloc(&dx, sizeof(int) * 4096));
    CK(hipMalloc(&dc, sizeof(int) * 4096));
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    // one-block-per-CU kernels (140 KB of LDS each): 224 + 32 blocks fit the chip only if the masks are disjoint CU sets;
    // launched big-first: without working masks the 32-block kernel would queue behind the big one
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    hipLaunchKernelGGL(fat_kernel, dim3(32), dim3(1024), 140 * 1024, s0, 1000L);   // warm-up: code object, queues
    hipLaunchKernelGGL(fat_kernel, dim3(224), dim3(1024), 140 * 1024, s1, 1000L);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a0, s0));
      hipLaunchKernelGGL(fat_kernel, dim3(32), dim3(1024), 140 * 1024, s0, 200000L);
      CK(hipEventRecord(a1, s0));
      CK(hipDeviceSynchronize());
      float t;
      CK(hipEventElapsedTime(&t, a0, a1));
      printf("alone: masked-32 kernel (2 ms spin) %.3f ms\n", t);
    }
    CK(hipEventRecord(b0, s1));
    hipLaunchKernelGGL(fat_kernel, dim3(224), dim3(1024), 140 * 1024, s1, 300000L);   // one resident round, 3 ms, 224 CUs
    CK(hipEventRecord(b1, s1));
    CK(hipEventRecord(a0, s0));
    hipLaunchKernelGGL(fat_kernel, dim3(32), dim3(1024), 140 * 1024, s0, 200000L);        // 2 ms on 32 CUs
    CK(hipEventRecord(a1, s0));
    CK(hipDeviceSynchronize());
    float ta, tb, tab;
    CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&tb, b0, b1)); CK(hipEventElapsedTime(&tab, a0, b1));
    printf("concurrent: masked-32 kernel %.3f ms, masked-224 kernel %.3f ms, first start -> last end %.3f ms\n", ta, tb, tab);
  }
  return 0;
}
