// Developer probe 12: operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of D[4][4] += A[4][1] * B[1][4]).
// Prints, for a few lanes, which (block, row, column) the A, B and D registers belong to.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out)
{
    const int l = threadIdx.x;
    // A = 1000 + lane, B = 1 in exactly one lane at a time is too slow; use structured values:
    // a(lane) = (lane >> 2) * 100 + (lane & 3) + 1   -> encodes (block, i)
    // b(lane) = 1                                      -> D[i][j] = a(block, i) for all j  => tells which a a D register sees
    float a = (float)((l >> 2) * 100 + (l & 3) + 1), b = 1.f;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
    // second experiment: a = 1, b encodes (block, j)
    float a2 = 1.f, b2 = (float)((l >> 2) * 100 + (l & 3) + 1);
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    e = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, b2, e, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = e[r];
}
int main()
{
    float* d; (void)hipMalloc(&d, 512 * 4);
    probe<<<1, 64>>>(d);
    float h[512]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 3, 4, 5, 21, 63}) {
        printf("lane %2d: D(a-coded) = %g %g %g %g   D(b-coded) = %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3],
               h[256 + l * 4], h[256 + l * 4 + 1], h[256 + l * 4 + 2], h[256 + l * 4 + 3]);
    }
    return 0;
}
