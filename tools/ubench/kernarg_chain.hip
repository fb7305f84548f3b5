// Does the hardware kernarg preload (first dwords of the kernel arguments in SGPRs at wave start) shorten the dependent chain
// at the head of a block?  Chain A: 300-byte struct by value -> s_load of its fields -> dependent s_load of a table word in
// device memory -> dependent vector load -> store.  Chain B: the pointers as leading scalar arguments (preloaded with
// -mllvm -amdgpu-kernarg-preload-count=8): table word and data load start without a kernarg round trip.  300 dependent launches.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Args { const int* tbl; const int* data; int* out; int pad[69]; };
__global__ __launch_bounds__(256) void chain_struct(Args a) {
  const int t = a.tbl[blockIdx.x & 3];                  // scalar load (uniform)
  const int v = a.data[t + threadIdx.x];
  if (v == 0x7fffffff) a.out[blockIdx.x] = v + a.pad[5];
}
__global__ __launch_bounds__(256) void chain_scalar(const int* __restrict__ tbl, const int* __restrict__ data, int* __restrict__ out, int p5) {
  const int t = tbl[blockIdx.x & 3];
  const int v = data[t + threadIdx.x];
  if (v == 0x7fffffff) out[blockIdx.x] = v + p5;
}
__global__ __launch_bounds__(256) void empty_k(int* out) { if (threadIdx.x == 12345) out[0] = 1; }
int main() {
  int *tbl, *data, *out;
  (void)hipMalloc(&tbl, 64); (void)hipMalloc(&data, 1 << 20); (void)hipMalloc(&out, 1 << 16);
  (void)hipMemset(tbl, 0, 64); (void)hipMemset(data, 0, 1 << 20);
  Args a{}; a.tbl = tbl; a.data = data; a.out = out;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grid : {16, 400, 1568}) {
    float ms[3];
    for (int v = 0; v < 3; v++) {
      for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 300; i++) {
          if (v == 0) chain_struct<<<grid, 256>>>(a);
          else if (v == 1) chain_scalar<<<grid, 256>>>(tbl, data, out, 5);
          else empty_k<<<grid, 256>>>(out);
        }
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms[v], e0, e1);
      }
    }
    printf("grid %5d: struct kernarg chain %.2f us | leading scalar (preloaded) arguments %.2f us | empty kernel %.2f us   per launch\n", grid,
           ms[0] * 1e3 / 300, ms[1] * 1e3 / 300, ms[2] * 1e3 / 300);
  }
  return 0;
}
