// HBM read/write rates of the access shapes the kernels of this library use (MI355X).
//   build: hipcc --offload-arch=gfx950 -O3 tools/probe/hbm_probe.hip -o tools/probe/hbm_probe
// Every test moves the same [rows][pitch bytes] region; what differs is which bytes a wave touches per instruction
// and in which order a workgroup walks its rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// (a) linear: lane reads 16 contiguous bytes, a wave 1 KB, consecutive waves consecutive KB (grid-stride)
__global__ void read_linear(const u32x4* __restrict__ p, size_t n16, unsigned* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
// (b) product shape: workgroup b owns rows [R b/G, R (b+1)/G); it walks the row width in stages of `piece` bytes:
//     per stage every row contributes `piece` contiguous bytes (8 lanes x 16 B = 128 B when piece = 128)
__global__ void read_staged(const char* __restrict__ p, long rows, long pitch, int piece, unsigned* sink) {
  const long r0 = rows * blockIdx.x / gridDim.x, r1 = rows * (blockIdx.x + 1) / gridDim.x;
  const int lanes_per_row = piece / 16, rows_per_pass = blockDim.x / lanes_per_row;
  const int lr = threadIdx.x / lanes_per_row, lc = threadIdx.x % lanes_per_row;
  u32x4 acc = {0, 0, 0, 0};
  for (long k = 0; k < pitch; k += piece)
    for (long r = r0 + lr; r < r1; r += rows_per_pass) acc ^= *(const u32x4*)(p + r * pitch + k + 16 * lc);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
// (c) row shape: one wave per row, the whole row contiguous (LayerNorm kernels)
__global__ void read_rows(const char* __restrict__ p, long rows, long pitch, unsigned* sink) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, wpb = blockDim.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  for (long r = blockIdx.x * (long)wpb + w; r < rows; r += (long)gridDim.x * wpb)
    for (long k = 16 * l; k < pitch; k += 1024) acc ^= *(const u32x4*)(p + r * pitch + k);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
// writes: linear, and lane = row with `piece` contiguous bytes per group of piece/16 lanes
__global__ void write_linear(u32x4* __restrict__ p, size_t n16) {
  const u32x4 v = {1, 2, 3, 4};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void write_pieces(char* __restrict__ p, long rows, long pitch, int piece) {
  // a wave instruction writes 64 lanes x 16 B as (1024 / piece) rows x piece bytes; the workgroup walks the row
  // width 64 columns (128 B) at a time like an output chunk of the K=200 product
  const int lpr = piece / 16;
  const u32x4 v = {1, 2, 3, 4};
  const long r0 = rows * blockIdx.x / gridDim.x, r1 = rows * (blockIdx.x + 1) / gridDim.x;
  const int rows_per_pass = blockDim.x / lpr, lr = threadIdx.x / lpr, lc = threadIdx.x % lpr;
  for (long r = r0 + lr; r < r1; r += rows_per_pass)
    for (long k = 0; k < pitch; k += piece) *(u32x4*)(p + r * pitch + k + 16 * lc) = v;
}
template <class F> static double timeit(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a));
  const int n = 20;
  for (int i = 0; i < n; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / n * 1e-3;
}
int main() {
  const long rows = 50718;
  unsigned* sink; CK(hipMalloc(&sink, 4));
  for (long pitch : {1280l, 3840l}) {
    const size_t bytes = (size_t)rows * pitch;
    char* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes));
    printf("region %ld rows x %ld B = %.1f MB\n", rows, pitch, bytes / 1e6);
    for (int g : {256, 512, 1024, 2048}) {
      double t = timeit([&] { read_linear<<<g, 512>>>((const u32x4*)p, bytes / 16, sink); });
      printf("  read linear        grid %5d          %6.1f us  %5.2f TB/s\n", g, t * 1e6, bytes / t / 1e12);
    }
    for (int piece : {128, 256, 512, 1280})
      for (int g : {256, 512}) {
        double t = timeit([&] { read_staged<<<g, 448>>>(p, rows, pitch, piece > pitch ? (int)pitch : piece, sink); });
        printf("  read staged piece %4d grid %4d       %6.1f us  %5.2f TB/s\n", piece, g, t * 1e6, bytes / t / 1e12);
      }
    for (int g : {1024, 4096}) {
      double t = timeit([&] { read_rows<<<g, 256>>>(p, rows, pitch, sink); });
      printf("  read one wave/row  grid %5d          %6.1f us  %5.2f TB/s\n", g, t * 1e6, bytes / t / 1e12);
    }
    for (int g : {256, 1024}) {
      double t = timeit([&] { write_linear<<<g, 512>>>((u32x4*)p, bytes / 16); });
      printf("  write linear       grid %5d          %6.1f us  %5.2f TB/s\n", g, t * 1e6, bytes / t / 1e12);
    }
    for (int piece : {16, 32, 64, 128, 256})
      for (int g : {512}) {
        double t = timeit([&] { write_pieces<<<g, 256>>>(p, rows, pitch, piece); });
        printf("  write pieces %4d B grid %4d          %6.1f us  %5.2f TB/s\n", piece, g, t * 1e6, bytes / t / 1e12);
      }
    CK(hipFree(p));
  }
  return 0;
}
