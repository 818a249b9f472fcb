// How fast is the layer-1 operand gather of the convolution forward in the LDS pipeline, per form?  One workgroup of
// four waves per CU reads a staged [4][3][30][30] frame group the way conv1_tile does (lane = (row lr, k index lk),
// rows = consecutive output positions, stride 4) and sums what it reads:
//   b32    four ds_read_b32 per quad (k = 16 q + 4 u + lk): the shipped form
//   b128   one 16-byte read per quad (k = 16 q + 4 lk .. + 3): rows of odd ky are only 8-byte aligned
//   b64x2  two 8-byte reads per quad (same elements)
// Reported: ns per quad per wave (all waves busy).   hipcc --offload-arch=gfx950 -O3 tools/ldsprobe.hip -o tools/debug/ldsprobe
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int C = 3, H = 30, W = 30, CHW = C * H * W, G = 4, K1 = 192, Q1 = 12, S1 = 4, K = 8, W1o = 6, M1 = 36;
typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2 __attribute__((ext_vector_type(2), aligned(8)));

__device__ int patch_offset(int k) { const int c = k / 64, r = k % 64, ky = r / 8, kx = r % 8; return c * H * W + ky * W + kx; }

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int reps) {
    __shared__ __attribute__((aligned(16))) float img[G * CHW + 64];
    __shared__ int ko32[Q1 * 16], ko128[Q1 * 4];
    for (int i = threadIdx.x; i < G * CHW; i += 256) img[i] = (float)(i % 13);
    for (int i = threadIdx.x; i < Q1 * 16; i += 256) { const int u = i & 3, lk = (i >> 2) & 3, q = i >> 4; ko32[i] = patch_offset(16 * q + 4 * u + lk); }
    for (int i = threadIdx.x; i < Q1 * 4; i += 256) { const int lk = i & 3, q = i >> 2; ko128[i] = patch_offset(16 * q + 4 * lk); }
    __syncthreads();
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        for (int t = wave; t < 8; t += 4) {
            const int row = t * 16 + lr, im = row / M1, pos = row % M1, oy = pos / W1o, ox = pos % W1o;
            const float* base = img + im * CHW + S1 * oy * W + S1 * ox;
            for (int q = 0; q < Q1; ++q) {
                if (MODE == 0) {
                    const int4 ko = reinterpret_cast<const int4*>(ko32)[q * 4 + lk];
                    acc += base[ko.x] + base[ko.y] + base[ko.z] + base[ko.w];
                } else if (MODE == 1) {
                    const f4 v = *reinterpret_cast<const f4*>(base + ko128[q * 4 + lk]);
                    acc += v.x + v.y + v.z + v.w;
                } else {
                    const float* p = base + ko128[q * 4 + lk];
                    const f2 a = *reinterpret_cast<const f2*>(p), b = *reinterpret_cast<const f2*>(p + 2);
                    acc += a.x + a.y + b.x + b.y;
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    float* out; hipMalloc(&out, 512 * 256 * 4);
    const int reps = 200;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it = 0; it < 2; ++it) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(512), dim3(256), 0, 0, out, reps);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(512), dim3(256), 0, 0, out, reps);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(512), dim3(256), 0, 0, out, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float h; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
        printf("%s: %.1f us per launch, %.1f ns per quad per wave (sum %g)\n", mode == 0 ? "b32  " : mode == 1 ? "b128 " : "b64x2",
               ms * 1000, ms * 1e6 / (reps * 2.0 * Q1), h);
    }
    return 0;
}
