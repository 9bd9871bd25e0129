// What do gfx950's fp8 conversion instructions do?  Decodes all 256 byte values with v_cvt_pk_f32_fp8 and encodes a
// list of floats with v_cvt_pk_fp8_f32 (rounding, saturation, NaN).  Build: hipcc --offload-arch=gfx950 -O2 tools/fp8_probe.hip -o tools/fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* in, float* out, const float* fin, unsigned* bits) {
    unsigned w = in[threadIdx.x];
    f2 a = __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
    f2 b = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
    out[threadIdx.x * 4 + 0] = a.x; out[threadIdx.x * 4 + 1] = a.y; out[threadIdx.x * 4 + 2] = b.x; out[threadIdx.x * 4 + 3] = b.y;
    float x = fin[threadIdx.x * 2], y = fin[threadIdx.x * 2 + 1];
    bits[threadIdx.x] = __builtin_amdgcn_cvt_pk_fp8_f32(x, y, 0, false);
}
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return 1; } } while (0)
int main() {
    unsigned h_in[64]; for (int i = 0; i < 64; ++i) h_in[i] = (4*i) | ((4*i+1) << 8) | ((4*i+2) << 16) | ((4*i+3) << 24);
    float h_f[128] = {0.0f, 1.0f, 448.f, 449.f, 500.f, 1e9f, -448.f, -1e9f, 0.001953125f, 0.0009765625f, 0.3f, 0.0625f, 17.f, 18.f, 19.f, 240.f,
                      2.5f, 3.5f, 1.0625f, 1.1875f, 464.f, 480.f, NAN, INFINITY, 0.00146484375f, 0.0029296875f, 447.9f, 432.f, 1e-9f, -0.f};
    unsigned *d_in, *d_bits; float *d_out, *d_f;
    CK(hipMalloc(&d_in, 256)); CK(hipMalloc(&d_out, 1024)); CK(hipMalloc(&d_f, 512)); CK(hipMalloc(&d_bits, 256));
    CK(hipMemcpy(d_in, h_in, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(d_f, h_f, 512, hipMemcpyHostToDevice));
    k<<<1, 64>>>(d_in, d_out, d_f, d_bits);
    float h_out[256]; unsigned h_bits[64];
    CK(hipMemcpy(h_out, d_out, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_bits, d_bits, 256, hipMemcpyDeviceToHost));
    for (int i = 0; i < 256; ++i) printf("dec %02x %.9g\n", i, h_out[i]);
    for (int i = 0; i < 15; ++i) printf("enc %.9g %.9g -> %02x %02x\n", h_f[2*i], h_f[2*i+1], h_bits[i] & 0xff, (h_bits[i] >> 8) & 0xff);
    return 0;
}
