// Probe (gfx950): what does an out-of-range lane of `buffer_load_dwordx4 ... lds` leave in LDS -- zeros or the old bytes?
// build: hipcc --offload-arch=gfx950 -O2 tools/debug/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* src, int bytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4];
    const int t = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[t * 4 + i] = 7.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
    const int voff = (t < 32) ? t * 16 : (int)0xFFFFFFF0u;      // lanes 32..63 out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[t * 4 + i] = lds[t * 4 + i];
}
int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
    float *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 512, o);
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("lane 0: %g %g %g %g | lane 31: %g %g | lane 32 (OOB): %g %g %g %g | lane 63 (OOB): %g\n", r[0], r[1], r[2], r[3], r[124], r[125],
           r[128], r[129], r[130], r[131], r[252]);
    return 0;
}
