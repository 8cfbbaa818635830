// scratch: determine the operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 and its cbsz/abid broadcast
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CBSZ, int ABID>
__global__ void k(const double* a, const double* b, double* d) {
  int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, CBSZ, ABID, 0);
}
int main() {
  double ha[64], hb[64], hd[64], *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  for (int variant = 0; variant < 3; ++variant) {
    printf("variant %d\n", variant);
    for (int la = 0; la < 64; ++la) {
      for (int l = 0; l < 64; ++l) { ha[l] = (l == la); hb[l] = l + 1; }
      hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
      if (variant == 0) k<0, 0><<<1, 64>>>(da, db, dd);
      if (variant == 1) k<2, 1><<<1, 64>>>(da, db, dd);
      if (variant == 2) k<1, 1><<<1, 64>>>(da, db, dd);
      hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
      printf("la=%2d:", la);
      for (int l = 0; l < 64; ++l) if (hd[l] != 0) printf(" d[%d]=b[%d]", l, (int)hd[l] - 1);
      printf("\n");
    }
  }
  return 0;
}
