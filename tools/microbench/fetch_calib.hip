// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the stride-2 conv's window DMA:
// MI355X_MICROARCH.md says FETCH_SIZE reports exactly 1/2 of the bytes of a WIDE coalesced read (128-byte requests tallied
// at 64 B) and that other widths are uncalibrated.  Each kernel sums what it reads (global_load_dwordx4 per lane) from a
// 1.06 GB buffer of 144-byte records (7,372,800 of them: one B=8 x 720p activation tensor), far beyond the Infinity Cache:
//   full      every byte, lane-linear                                  -> 1061.7 MB requested
//   q64       bytes [0,64) of every record  (4 lanes per record)       ->  471.9 MB requested
//   q64_hi    bytes [64,128) of every record                           ->  471.9 MB
//   h128      bytes [0,128) of every record (8 lanes per record)       ->  943.7 MB
// Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int LANES_PER_REC, int BYTE0>
__global__ __launch_bounds__(256) void read_part(const char *__restrict__ src, size_t nrec, unsigned *sink)
{
    unsigned acc = 0;
    const size_t total = nrec * LANES_PER_REC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t rec = i / LANES_PER_REC;
        const int piece = (int)(i - rec * LANES_PER_REC);
        const u32x4 v = *reinterpret_cast<const u32x4 *>(src + rec * 144 + BYTE0 + piece * 16);
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void read_full(const char *__restrict__ src, size_t n16, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(src + i * 16);
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t nrec = 7372800, bytes = nrec * 144;
    char *buf; unsigned *sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        read_full<<<256 * 8, 256>>>(buf, bytes / 16, sink);
        read_part<4, 0><<<256 * 8, 256>>>(buf, nrec, sink);
        read_part<4, 64><<<256 * 8, 256>>>(buf, nrec, sink);
        read_part<8, 0><<<256 * 8, 256>>>(buf, nrec, sink);
    }
    CHECK(hipDeviceSynchronize());
    printf("requested MB: full %.1f  q64 %.1f  q64_hi %.1f  h128 %.1f\n", bytes / 1e6, nrec * 64 / 1e6, nrec * 64 / 1e6, nrec * 128 / 1e6);
    return 0;
}
