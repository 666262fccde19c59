#include <hip/hip_runtime.h>
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* in, float* out) {
    unsigned w = in[threadIdx.x];
    f32x2_t a = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 0);
    f32x2_t b = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 1);
    f32x2_t c = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 2);
    f32x2_t d = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 3);
    out[8*threadIdx.x+0]=a.x; out[8*threadIdx.x+1]=a.y; out[8*threadIdx.x+2]=b.x; out[8*threadIdx.x+3]=b.y;
    out[8*threadIdx.x+4]=c.x; out[8*threadIdx.x+5]=c.y; out[8*threadIdx.x+6]=d.x; out[8*threadIdx.x+7]=d.y;
}
int main(){ unsigned h[64]; for(int i=0;i<64;i++) h[i]=0x32103210u ^ (i==1?0x01230123u:0); unsigned* di; float* dout; hipMalloc(&di,256); hipMalloc(&dout,64*8*4); hipMemcpy(di,h,256,hipMemcpyHostToDevice); hipLaunchKernelGGL(k,1,64,0,0,di,dout); float o[16]; hipMemcpy(o,dout,64,hipMemcpyDeviceToHost); for(int i=0;i<8;i++) printf("%g ",o[i]); printf("\n"); return 0; }
