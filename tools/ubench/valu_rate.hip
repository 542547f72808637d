// valu_rate.hip -- issue rate of the integer VALU ops the pair kernels lean on (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int OP>
__global__ void k(unsigned* out, int iters, unsigned seed) {
	unsigned a[8], b = seed + threadIdx.x, c = seed * 3 + 1;
	for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
	unsigned long long w[4] = {b, c, b + 1ull, c + 2ull};
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < REP; r++) {
			unsigned& x = a[r & 7];      // 8 independent chains
			if (OP == 0) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 1) asm volatile("v_sad_u32 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(c), "v"(x));
			if (OP == 2) asm volatile("v_sad_u16 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(c), "v"(x));
			if (OP == 3) asm volatile("v_sad_u8 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(c), "v"(x));
			if (OP == 4) asm volatile("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(c), "v"(x));
			if (OP == 5) asm volatile("v_dot4_u32_u8 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(c), "v"(x));
			if (OP == 6) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(c), "v"(x));
			if (OP == 7) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(c), "v"(x));
			if (OP == 8) asm volatile("v_max_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 9) asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(x) : "v"(b), "v"(x));
			if (OP == 10) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 11) asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 12) asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(c), "v"(x));
			if (OP == 13) asm volatile("v_add_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 15) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(b));
			if (OP == 16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(b));
			if (OP == 17) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(x), "v"(c));
			if (OP == 18) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
			if (OP == 19) asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 20) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(x) : "v"(x));
			if (OP == 21) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 22) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(c), "v"(x));
			if (OP == 23) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 24) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 25) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 26) asm volatile("v_min_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 27) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(b));
			if (OP == 28) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(b), "v"(c));
			if (OP == 29) asm volatile("v_sad_u8 %0, %1, %2, %3" : "=v"(x) : "v"(b), "v"(c), "v"(b));
			if (OP == 30) { unsigned long long& y = w[r & 3]; unsigned long long cc; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(y), "=s"(cc) : "v"(b), "v"(c), "v"(y)); }
			if (OP == 31) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 32) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 33) asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 34) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(x), "v"(b));
			if (OP == 35) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
			if (OP == 36) { unsigned long long& y = w[r & 3]; asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(y) : "v"(y), "v"(w[(r + 1) & 3])); }
			if (OP == 37) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[10:11]" : "=v"(x) : "v"(x), "v"(b) : "s10", "s11");
			if (OP == 38) asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n\tv_add_u32 %3, %3, %2" : "=v"(x), "+v"(c) : "v"(x), "v"(b));      // two instructions
			if (OP == 39) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(x), "v"(b) : "vcc");      // two instructions
			if (OP == 40) asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(x) : "v"(x), "v"(b) : "vcc");
			if (OP == 41) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(b), "v"(c));      // no dependency
			if (OP == 42) asm volatile("v_cmp_lt_u32_e64 s[10:11], %1, %2\n\tv_cndmask_b32_e64 %0, %1, %2, s[10:11]" : "=v"(x) : "v"(x), "v"(b) : "s10", "s11");
			if (OP == 14) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(x) : "v"(b), "v"(x));
		}
	}
	unsigned s = 0;
	for (int i = 0; i < 8; i++) s += a[i];
	for (int i = 0; i < 4; i++) s += (unsigned)w[i] + (unsigned)(w[i] >> 32);
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, unsigned* d) {
	const int iters = 2000, blocks = 256 * 8, threads = 256;     // 8 waves per SIMD
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<OP><<<blocks, threads>>>(d, 10, 1); hipDeviceSynchronize();
	hipEventRecord(e0); k<OP><<<blocks, threads>>>(d, iters, 1); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double insts_per_simd = (double)iters * REP * (blocks * threads / 64) / (256.0 * 4);
	printf("%-16s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}
int main() {
	unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
	run<0>("v_add_u32", d); run<1>("v_sad_u32", d); run<2>("v_sad_u16", d); run<3>("v_sad_u8", d); run<4>("v_dot2_u32_u16", d);
	run<5>("v_dot4_u32_u8", d); run<6>("v_mad_u32_u24", d); run<7>("v_add3_u32", d); run<8>("v_max_u32", d); run<9>("v_lshl_or_b32", d);
	run<10>("v_pk_add_u16", d); run<11>("v_pk_max_u16", d); run<12>("v_pk_mad_u16", d); run<13>("v_add_u32_dpp", d); run<14>("v_add_u32_sdwa", d);
	run<15>("v_permlane32_swap", d); run<16>("v_permlane16_swap", d); run<17>("v_perm_b32", d); run<18>("v_mov_b32_dpp", d); run<19>("v_and_b32", d);
	run<20>("v_lshlrev_b32", d); run<21>("v_sub_u32", d); run<22>("v_fma_f32", d); run<23>("v_add_f32", d); run<24>("v_mul_u32_u24", d);
	run<25>("v_xor_b32", d); run<26>("v_min_u32", d); run<27>("v_mov_b32", d); run<28>("v_add_u32 (no dep)", d); run<29>("v_sad_u8 (no dep)", d);
	run<30>("v_mad_u64_u32", d); run<31>("v_mul_lo_u32", d); run<32>("v_mul_hi_u32", d); run<33>("v_mul_hi_u32_u24", d); run<34>("v_cndmask_b32", d);
	run<35>("v_cmp_lt_u32", d); run<36>("v_lshl_add_u64", d);
	run<37>("v_cndmask e64 sgpr", d); run<38>("cndmask+add (2)", d); run<39>("cmp+cndmask (2)", d); run<40>("v_addc_co_u32", d); run<41>("v_cndmask (no dep)", d);
	run<42>("cmp+cndmask sgpr(2)", d);
	return 0;
}
