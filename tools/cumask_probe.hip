// Which CUs does a CU-masked stream use?  hipExtStreamCreateWithCUMask(mask) -> launch many workgroups -> each records
// (XCC_ID, SE_ID, CU_ID) from the hardware id registers -> histogram.  usage: cumask_probe <hex words of the mask, low first>
// build: hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/bin/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void who(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // burn some time so that many workgroups are resident at once
    float x = threadIdx.x;
    for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hw + (x < 0 ? 1 : 0);
    }
}

int main(int argc, char** argv) {
    std::vector<uint32_t> mask;
    for (int i = 1; i < argc; ++i) mask.push_back((uint32_t)strtoul(argv[i], nullptr, 16));
    hipStream_t st = nullptr;
    if (mask.empty()) { hipStreamCreate(&st); printf("no mask\n"); }
    else if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("create failed\n"); return 1; }
    const int n = 4096;
    unsigned* d;
    hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(who, dim3(n), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(n * 2);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> cnt;
    for (int i = 0; i < n; ++i) {
        const unsigned xcc = h[i * 2] & 0xf, hw = h[i * 2 + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;   // gfx9 HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
        cnt[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    printf("%zu distinct (xcc, se, sh, cu):", cnt.size());
    int k = 0;
    for (auto& kv : cnt) { if (k++ < 400) printf(" %x.%x.%x.%x", kv.first >> 12, (kv.first >> 8) & 0xf, (kv.first >> 4) & 0xf, kv.first & 0xf); }
    printf("\n");
    printf("first 48 workgroups -> xcc:");
    for (int i = 0; i < 48; ++i) printf(" %u", h[i * 2] & 0xf);
    printf("\n");
    uint32_t got[16] = {0};
    if (!mask.empty() && hipExtStreamGetCUMask(st, 16, got) == hipSuccess) { printf("stream mask:"); for (int i = 0; i < 8; ++i) printf(" %08x", got[i]); printf("\n"); }
    return 0;
}
