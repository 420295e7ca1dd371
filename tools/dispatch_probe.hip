// Where does the dispatcher put the workgroups of a launch that does not fill the chip? Blocks of 256 threads with a given
// dynamic LDS size record (XCC, SE, CU) and spin long enough to be co-resident; the histogram of blocks per CU is printed.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o tools/dispatch_probe.bin ; tools/dispatch_probe.bin <blocks> <lds bytes>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
    extern __shared__ float sm[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    sm[threadIdx.x] = x;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc | (sm[0] == 12345.f ? 0x80000000u : 0u);
    }
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 770;
    const int lds = argc > 2 ? atoi(argv[2]) : 36864;
    unsigned* d;
    hipMalloc(&d, blocks * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    probe<<<blocks, 256, lds>>>(d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu;
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("%d blocks, %d B LDS: %zu distinct CUs used;", blocks, lds, per_cu.size());
    for (auto& kv : hist) printf("  %d CUs hold %d blocks", kv.second, kv.first);
    printf("\n");
    return 0;
}
