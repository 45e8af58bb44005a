// tr_read_probe.hip -- pins the data movement of gfx950's LDS transpose read (ds_read_b64_tr_b16) before a kernel is built on it.
// LDS holds lds[i] = i (16-bit elements).  Every 16-lane group g reads a block of 4 rows; lane i of the group supplies the address of
// the i-th 8-byte chunk of the block in row-major order (row = i / 4, chunk = i % 4) for a caller-chosen row stride.  Prints, per
// lane, the 4 element indices it received.  Expected (cdna_hip_programming.md): lane i receives column i of the 4 x 16 block,
// elements [row 0..3][col i].
// build + run: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
#define LDSP(T) __attribute__((address_space(3))) T

__global__ void probe(unsigned short* out, int rowstride_bytes, int shift_rows) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    const int byte_off = (g * 4 + shift_rows) * rowstride_bytes + (i >> 2) * rowstride_bytes + (i & 3) * 8;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP(v4s)*)((LDSP(char)*)lds + byte_off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}

int main() {
    unsigned short* d;
    unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    const int cases[3][2] = {{32, 0}, {64, 0}, {128, 3}};
    int bad = 0;
    for (auto& c : cases) {
        probe<<<1, 64>>>(d, c[0], c[1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d bytes, first row %d:\n", c[0], c[1]);
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, i = l & 15;
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) {
                const int want = ((g * 4 + c[1] + j) * c[0]) / 2 + i;  // element [row j][col i] of the group's block
                printf(" %5d%s", h[l * 4 + j], h[l * 4 + j] == want ? "" : "!");
                bad += h[l * 4 + j] != want;
            }
            printf("\n");
        }
    }
    printf("%s\n", bad ? "MISMATCH against the expected transpose" : "transpose read as expected: lane i <- column i of the 4 x 16 block");
    return bad != 0;
}
