// Probe of ds_read_b64_tr_b16 and global_load_lds on gfx950: what each lane receives.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/tr16.hip -o tools/probe/tr16 && tools/probe/tr16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

// LDS holds u16 value = its own element index (address / 2).  Lane l supplies byte address addr[l]; out[l][j] = element j it got.
__global__ void k_tr(const int* addr, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)((char*)lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
// glds: lane l copies 16 bytes from src[idx[l]] to LDS chunk base + 16*l (wave-uniform base); inactive lanes?
__global__ void k_glds(const uint4* src, const int* idx, uint32_t* out, int active) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    if ((int)threadIdx.x < active)
        __builtin_amdgcn_global_load_lds((glb_ptr)(src + idx[threadIdx.x]), (lds_ptr)(lds + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    int h_addr[64];
    // canonical: lane i of a 16-group supplies row (i>>2), piece (i&3) of a [4][16] u16 block with row stride RS elements
    const int RS = 40;   // elements; arbitrary (80 bytes)
    for (int l = 0; l < 64; ++l) {
        const int grp = l >> 4, i = l & 15;
        h_addr[l] = 2 * (grp * 1024 + (i >> 2) * RS + (i & 3) * 4);
    }
    int* d_addr; uint16_t* d_out; std::vector<uint16_t> h_out(256);
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, 512);
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    k_tr<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(h_out.data(), d_out, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int grp = l >> 4, i = l & 15;
        for (int j = 0; j < 4; ++j) {
            const int want = grp * 1024 + j * RS + i;       // column i of row j
            if (h_out[l * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, j, h_out[l * 4 + j], want); ++bad; }
        }
    }
    printf("tr16 model 'lane i of a 16-group gets column i, rows 0..3 of the [4][16] block its group's lanes point at': %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    if (bad) for (int l = 0; l < 20; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);

    // glds
    std::vector<uint32_t> h_src(4 * 256); for (int i = 0; i < 1024; ++i) h_src[i] = i;
    int h_idx[64]; for (int l = 0; l < 64; ++l) h_idx[l] = (l * 7) % 200;     // arbitrary per-lane sources
    uint4* d_src; int* d_idx; uint32_t* d_o; std::vector<uint32_t> h_o(1024);
    hipMalloc(&d_src, 4096); hipMalloc(&d_idx, 256); hipMalloc(&d_o, 4096);
    hipMemcpy(d_src, h_src.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(d_idx, h_idx, 256, hipMemcpyHostToDevice);
    for (int active : {64, 40}) {
        k_glds<<<1, 64>>>(d_src, d_idx, d_o, active);
        hipMemcpy(h_o.data(), d_o, 4096, hipMemcpyDeviceToHost);
        int b2 = 0, touched_outside = 0;
        for (int i = 0; i < 1024; ++i) {
            const int rel = i - 256;
            if (rel >= 0 && rel < 256) {
                const int l = rel / 4, want = (l < active) ? (int)(h_idx[l] * 4 + rel % 4) : (int)0xdeadbeef;
                if ((int)h_o[i] != want) { if (b2 < 6) printf("glds active %d: lds dword %d (lane %d) = %x want %x\n", active, i, l, h_o[i], want); ++b2; }
            } else if (h_o[i] != 0xdeadbeefu) ++touched_outside;
        }
        printf("glds (active lanes %d): per-lane source, dest = base + 16*lane, inactive lanes leave LDS alone: %s (%d mismatches, %d outside)\n",
               active, (b2 || touched_outside) ? "WRONG" : "ok", b2, touched_outside);
    }
    return 0;
}
