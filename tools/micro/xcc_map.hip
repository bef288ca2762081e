// tools/micro/xcc_map.hip -- which XCD does workgroup b of a launch run on, and is that the same from launch to launch?
// hipcc --offload-arch=gfx950 -O2 tools/micro/xcc_map.hip -o /tmp/xcc_map && /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void k(int *out)
{
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x] = (int)(id & 0xf);
    }
}
int main()
{
    const int G = 216, L = 8;
    int *d; hipMalloc(&d, G * L * sizeof(int));
    for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k, dim3(G), dim3(1024), 65536, 0, d + l * G);
    hipDeviceSynchronize();
    std::vector<int> h(G * L); hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    int same = 0, rr = 0;
    for (int b = 0; b < G; ++b) { bool s = true; for (int l = 1; l < L; ++l) s = s && h[l * G + b] == h[b]; same += s; }
    for (int l = 0; l < L; ++l) { int ok = 0; for (int b = 0; b < G; ++b) ok += h[l * G + b] == (h[l * G] + b) % 8; printf("launch %d: block 0 on XCD %d, round-robin holds for %d of %d blocks\n", l, h[l * G], ok, G); }
    printf("blocks on the same XCD in all %d launches: %d of %d\n", L, same, G);
    return 0;
}
