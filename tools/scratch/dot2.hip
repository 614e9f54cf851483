#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out) {
    h2 a = {(_Float16)1.5f, (_Float16)2.0f}, b = {(_Float16)4.0f, (_Float16)0.25f};
    out[0] = __builtin_amdgcn_fdot2(a, b, 10.0f, false);                 // 10 + 6 + 0.5 = 16.5
    h2 t = {(_Float16)3e-6f, (_Float16)0.0f}, u = {(_Float16)1024.0f, (_Float16)0.0f};
    out[1] = __builtin_amdgcn_fdot2(t, u, 0.0f, false);                  // subnormal * 1024
    out[2] = (float)t[0] * 1024.0f;
    b2 c = {(__bf16)1.5f, (__bf16)2.0f}, d = {(__bf16)4.0f, (__bf16)0.25f};
    out[3] = __builtin_amdgcn_fdot2_f32_bf16(c, d, 10.0f, false);
    h2 e = {(_Float16)0.1f, (_Float16)0.3f}, f = {(_Float16)0.7f, (_Float16)0.9f};
    out[4] = __builtin_amdgcn_fdot2(e, f, 0.0f, false);
    out[5] = fmaf((float)e[1], (float)f[1], fmaf((float)e[0], (float)f[0], 0.0f));
}
int main() {
    float* d; hipMalloc(&d, 64); k<<<1,1>>>(d); float h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 6; i++) printf("out[%d] = %.9g\n", i, h[i]);
    return 0;
}
