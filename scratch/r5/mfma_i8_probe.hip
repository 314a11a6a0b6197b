// Round 5 probe: operand / result mapping of v_mfma_i32_16x16x64_i8 on gfx950, as the stencil kernels assume it:
//   lane l = (g = l >> 4, t = l & 15) holds 16 bytes; A lane: row m = t, bytes = 16 K-elements of K-group g; B lane: column n = t, same K-group;
//   D: lane (n = l & 15, q = l >> 4), register r -> row m = 4 q + r.
// Only the CONSISTENCY of the (g, byte) -> k mapping between A and B matters to the kernels, not the k numbering itself; the probe checks
// D[m][n] = sum_{g, i} A[m][g][i] * B[n][g][i] with random operands, then reports which byte of B pairs with byte i of A (identity expected).
//   hipcc --offload-arch=gfx950 -O2 scratch/r5/mfma_i8_probe.hip -o scratch/r5/mfma_i8_probe && gpurun -- scratch/r5/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) int i32x4;

__global__ void k(const i32x4* a, const i32x4* b, i32x4* d) {
  const int l = threadIdx.x;
  i32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[l], b[l], acc, 0, 0, 0);
  d[l] = acc;
}

int main() {
  int8_t A[64][16], B[64][16];
  int D[64][4];
  i32x4 *da, *db, *dd;
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
  srand(1);
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) { A[l][i] = (int8_t)(rand() % 255 - 127); B[l][i] = (int8_t)(rand() % 255 - 127); }
  hipMemcpy(da, A, 1024, hipMemcpyHostToDevice); hipMemcpy(db, B, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(D, dd, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int n = 0; n < 16; ++n) for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) {
    const int m = 4 * q + r;
    long s = 0;
    for (int g = 0; g < 4; ++g) for (int i = 0; i < 16; ++i) s += (int)A[g * 16 + m][i] * (int)B[g * 16 + n][i];
    if (s != D[q * 16 + n][r]) ++bad;
  }
  printf("assumed mapping: %d of 256 results differ\n", bad);
  // one-hot: A row 3, K-group 2, byte 5 = 1; B column 7: every (g, i) byte = 16 g + i  -> D[3][7] names the paired B byte
  for (int ga = 0; ga < 4; ++ga) for (int ia = 0; ia < 16; ia += 5) {
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) { A[l][i] = 0; B[l][i] = (int8_t)((l >> 4) * 16 + i); }
    A[ga * 16 + 3][ia] = 1;
    hipMemcpy(da, A, 1024, hipMemcpyHostToDevice); hipMemcpy(db, B, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(D, dd, 1024, hipMemcpyDeviceToHost);
    printf("A(g=%d, i=%2d) pairs with B byte code %3d (expected %3d); D[3][7] at lane %d reg %d\n", ga, ia, D[0 * 16 + 7][3], ga * 16 + ia, 7, 3);
  }
  return bad != 0;
}
