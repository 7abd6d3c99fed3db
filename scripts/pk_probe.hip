// GPU box: what v_pk_add_f32's op_sel / op_sel_hi / neg_lo / neg_hi select, on a = (1, 2), b = (10, 20).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* o) {
    f32x2 a = {1.f + threadIdx.x, 2.f}, b = {10.f, 20.f}, r;
#define T(i, MODS) asm volatile("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r) : "v"(a), "v"(b)); o[2 * i] = r.x; o[2 * i + 1] = r.y;
    T(0, "")
    T(1, "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]")
    T(2, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
    T(3, "neg_lo:[0,1] neg_hi:[1,0]")
    T(4, "op_sel:[1,0]")
    T(5, "op_sel_hi:[0,0]")
    T(6, "neg_lo:[1,0]")
    T(7, "op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]")
}
int main() {
    float* d; hipMalloc(&d, 64); k<<<1, 1>>>(d); float h[16]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    const char* n[] = {"default", "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]", "neg_lo:[0,1] neg_hi:[1,0]", "op_sel:[1,0]", "op_sel_hi:[0,0]", "neg_lo:[1,0]", "op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]"};
    for (int i = 0; i < 8; ++i) printf("%-45s -> (%g, %g)\n", n[i], h[2 * i], h[2 * i + 1]);
    return 0;
}
