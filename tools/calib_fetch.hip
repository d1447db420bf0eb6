// calib_fetch.hip — what FETCH_SIZE reports for the traversal kernels' access pattern (MI355X_MICROARCH.md, HBM section: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   k_stream : every lane reads consecutive float4s (the calibrated case: FETCH_SIZE = 1/2 of the bytes)
//   k_nodes  : every lane reads ONE 64-B node record the way HbmTree::planes does (3 x dwordx4 + 1 x dwordx2 from its own node),
//              nodes visited in a pseudo-random order, each exactly once; the table (1 GiB) is far larger than the Infinity Cache
// Known bytes: k_stream N x 16, k_nodes N x 64 (whole lines; 56 B of each are asked for).  Run under
//   rocprofv3 --pmc FETCH_SIZE -- tools/calib_fetch      and compare (scripts/r02_profiles.sh does, -> profiles/r02_fetch_calibration.txt).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct Node { float4 q0, q1, q2; int4 q3; };
__global__ void __launch_bounds__(256) k_stream(const float4* in, float4* out, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    for (; i < n; i += gridDim.x * 256) { float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_nodes(const Node* nodes, float4* out, uint32_t n_log2) {
    const uint32_t n = 1u << n_log2, mask = n - 1;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0; int c = 0;
    for (; i < n; i += gridDim.x * 256) {
        const uint32_t j = (i * 2654435761u + 12345u) & mask;          // odd multiplier: a permutation of [0, n)
        const Node* np = nodes + j;
        const float4 a = np->q0, b = np->q1, d = np->q2; const int2 ch = *(const int2*)&np->q3;
        acc += a.x + a.w + b.y + b.z + d.x + d.w; c += ch.x ^ ch.y;
    }
    if (acc == 123.456f && c == 77) out[0] = make_float4(acc, (float)c, 0, 0);
}
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
int main() {
    const uint32_t n_log2 = 24; const size_t n = 1u << n_log2;          // 16 M nodes = 1 GiB
    void *tab, *out;
    CK(hipMalloc(&tab, n * sizeof(Node))); CK(hipMalloc(&out, 256));
    CK(hipMemset(tab, 0, n * sizeof(Node)));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, (const float4*)tab, (float4*)out, (uint32_t)(n * 4));   // 1 GiB as float4s
    hipLaunchKernelGGL(k_nodes, dim3(256 * 8), dim3(256), 0, 0, (const Node*)tab, (float4*)out, n_log2);
    CK(hipDeviceSynchronize());
    printf("k_stream known bytes %zu\nk_nodes known bytes %zu (64-B lines; 56 B of each requested)\n", n * 64, n * 64);
    return 0;
}
