// What does ds_read_b64_tr_b16 (gfx950 LDS transpose read) return for a given set of per-lane addresses?
// LDS holds lds16[k] = k (16-bit elements); every lane passes a byte address and gets four 16-bit elements back; the probe prints, per
// lane, the element INDICES it received, for (a) the natural addresses lane*8 and (b) a [pixel][channel] tile with a 144-byte pixel
// pitch where lane (i = lane & 15, g = lane >> 4) points at pixel i/4 + 4*(g>>1), channels 4*(i%4) + 16*(g&1) .. +3.
// Expected (cdna_hip_programming.md): within a 16-lane group the lanes' 8-byte pieces form a 4 x 16 matrix (row r = pieces 4r..4r+3)
// and lane i receives column i: element j = piece (4j + i/4), 16-bit lane (i % 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(r_)); exit(1); } } while (0)

__global__ void tr_kernel(const unsigned* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds16[8192];
  for (int k = threadIdx.x; k < 8192; k += 64) lds16[k] = (unsigned short)k;
  __syncthreads();
  const unsigned a = addr[threadIdx.x] + (unsigned)(size_t)lds16;     // (the shared array sits at LDS offset 0 here; keep it general)
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}

static void run(const char* title, const unsigned* h_addr) {
  unsigned* d_addr; unsigned short* d_out; unsigned short h_out[256];
  CK(hipMalloc(&d_addr, 256)); CK(hipMalloc(&d_out, 512));
  CK(hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 0, 0, d_addr, d_out);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost));
  printf("%s\n", title);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, i = l & 15;
    printf("  lane %2d (addr %5u):", l, h_addr[l]);
    for (int j = 0; j < 4; ++j) {
      const unsigned expect = h_addr[16 * g + 4 * j + i / 4] / 2 + (i % 4);
      printf(" %5u%s", h_out[l * 4 + j], h_out[l * 4 + j] == expect ? "" : "!");
      bad += h_out[l * 4 + j] != expect;
    }
    printf("\n");
  }
  printf("  -> %s the expected gather (element j of lane i = 16-bit lane i%%4 of the piece of lane 4j + i/4 in its 16-lane group)\n", bad ? "DIFFERS FROM" : "matches");
  CK(hipFree(d_addr)); CK(hipFree(d_out));
}

int main() {
  unsigned a[64];
  for (int l = 0; l < 64; ++l) a[l] = l * 8;
  run("(a) natural addresses lane * 8", a);
  for (int l = 0; l < 64; ++l) {
    const int i = l & 15, g = l >> 4;
    a[l] = (unsigned)((i / 4 + 4 * (g >> 1)) * 144 + (4 * (i % 4) + 16 * (g & 1)) * 2);
  }
  run("(b) [pixel][channel] tile, 144-byte pixel pitch: lane (i, g) -> pixel i/4 + 4 (g>>1), channels 4 (i%4) + 16 (g&1)", a);
  return 0;
}
