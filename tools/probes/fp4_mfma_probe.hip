#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// one wave: A bits [32 rows][64 k], B bits [32 cols][64 k]; lane (r, h) holds 32 bits = k 32h .. 32h+31 of its row / column
__global__ void k_probe(const uint32_t* a_bits, const uint32_t* b_bits, float* out, int scaled) {
	const uint32_t lane = threadIdx.x, r = lane & 31, h = lane >> 5;
	const uint32_t wa = a_bits[r * 2 + h], wb = b_bits[r * 2 + h];
	v8i A = {0, 0, 0, 0, 0, 0, 0, 0}, B = {0, 0, 0, 0, 0, 0, 0, 0};
	A[0] = (int)((wa << 2) & 0x44444444u); A[1] = (int)(wa & 0x22222222u); A[2] = (int)((wa >> 2) & 0x11111111u); A[3] = (int)((wa >> 1) & 0x44444444u);
	B[0] = (int)(wb & 0x11111111u); B[1] = (int)(wb & 0x22222222u); B[2] = (int)(wb & 0x44444444u); B[3] = (int)((wb >> 3) & 0x11111111u);
	v16f acc;
	for (int i = 0; i < 16; i++) acc[i] = 0.f;
	if (scaled) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
	else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, 0, 0, 0);
	// D: col = lane & 31 (B's column), row = (reg & 3) + 8 (reg >> 2) + 4 h (A's row)
	for (int i = 0; i < 16; i++) out[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}
int main() {
	std::vector<uint32_t> a(64), b(64);
	srand(7);
	for (auto& x : a) x = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand();
	for (auto& x : b) x = (uint32_t)rand() * 2246822519u ^ (uint32_t)rand();
	uint32_t *da, *db; float* dout;
	hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 4096);
	hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
	for (int scaled = 0; scaled < 2; scaled++) {
		k_probe<<<1, 64>>>(da, db, dout, scaled);
		std::vector<float> o(1024);
		hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
		int bad = 0;
		for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
			const int want = __builtin_popcount(a[i * 2] & b[j * 2]) + __builtin_popcount(a[i * 2 + 1] & b[j * 2 + 1]);
			if (o[i * 32 + j] != (float)want) { if (bad < 5) printf("scaled %d (%d,%d): got %g want %d\n", scaled, i, j, o[i * 32 + j], want); bad++; }
		}
		printf("scaled %d: %d wrong of 1024\n", scaled, bad);
	}
	return 0;
}
