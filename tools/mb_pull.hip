// microbenchmark: what limits an affine trilinear pull at 256^3 on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
struct Aff { float m[12]; };
__device__ __forceinline__ float2 ld2(const float* p){ return *reinterpret_cast<const float2*>(p); }
template<int MODE, int CH>
__global__ void __launch_bounds__(256) kp(const float* __restrict__ src, int nx, int ny, int nz, Aff A, float* __restrict__ dst, int gx_, int gy_, int gz_) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.y*4 + w, i = blockIdx.z;
  const int kbase = blockIdx.x*64*CH;
  const float rx = fmaf(A.m[1],(float)j,A.m[0]*(float)i), ry = fmaf(A.m[5],(float)j,A.m[4]*(float)i), rz = fmaf(A.m[9],(float)j,A.m[8]*(float)i);
  float* row = dst + ((size_t)i*gy_ + j)*gz_;
  const unsigned nynz = ny*nz;
#pragma unroll
  for (int u=0;u<CH;++u){
    const int k = kbase + u*64 + lane;
    if (k >= gz_) continue;
    float gx = fmaf(A.m[2],(float)k,rx)+A.m[3], gy = fmaf(A.m[6],(float)k,ry)+A.m[7], gz = fmaf(A.m[10],(float)k,rz)+A.m[11];
    // clamp into interior so every mode is safe
    gx = fminf(fmaxf(gx, 0.f), (float)(nx-2)); gy = fminf(fmaxf(gy,0.f),(float)(ny-2)); gz = fminf(fmaxf(gz,0.f),(float)(nz-2));
    const float fx=floorf(gx), fy=floorf(gy), fz=floorf(gz);
    const float wx=gx-fx, wy=gy-fy, wz=gz-fz;
    const unsigned ix=(unsigned)(int)fx, iy=(unsigned)(int)fy, iz=(unsigned)(int)fz;
    unsigned off = __umul24(__umul24(ix,ny)+iy,nz)+iz;
    float out;
    if (MODE==0) { // dwordx2 unaligned, lerp
      const float2 a=ld2(src+off), b=ld2(src+off+nz), c=ld2(src+off+nynz), d=ld2(src+off+nynz+nz);
      const float a1=fmaf(wz,a.y-a.x,a.x), b1=fmaf(wz,b.y-b.x,b.x), c1=fmaf(wz,c.y-c.x,c.x), d1=fmaf(wz,d.y-d.x,d.x);
      const float ab=fmaf(wy,b1-a1,a1), cd=fmaf(wy,d1-c1,c1); out=fmaf(wx,cd-ab,ab);
    } else if (MODE==1) { // no loads
      out = wx+wy+wz+(float)off;
    } else if (MODE==2) { // 8 dword loads
      const float a0=src[off],a1_=src[off+1],b0=src[off+nz],b1_=src[off+nz+1],c0=src[off+nynz],c1_=src[off+nynz+1],d0=src[off+nynz+nz],d1_=src[off+nynz+nz+1];
      const float a1=fmaf(wz,a1_-a0,a0), b1=fmaf(wz,b1_-b0,b0), c1=fmaf(wz,c1_-c0,c0), d1=fmaf(wz,d1_-d0,d0);
      const float ab=fmaf(wy,b1-a1,a1), cd=fmaf(wy,d1-c1,c1); out=fmaf(wx,cd-ab,ab);
    } else if (MODE==3) { // single load (nearest)
      out = src[off]*wx;
    } else if (MODE==4) { // 4 dword loads (z0 only) 
      const float a0=src[off],b0=src[off+nz],c0=src[off+nynz],d0=src[off+nynz+nz];
      const float ab=fmaf(wy,b0-a0,a0), cd=fmaf(wy,d0-c0,c0); out=fmaf(wx,cd-ab,ab)+wz;
    } else { // MODE 5: 2 loads: row (x0,y0) dwordx2 and (x1,y1)
      const float2 a=ld2(src+off), d=ld2(src+off+nynz+nz);
      out = fmaf(wz,a.y-a.x,a.x)*wx + fmaf(wz,d.y-d.x,d.x)*wy;
    }
    row[k]=out;
  }
}
// plain copy for reference
__global__ void kcopy(const float4* __restrict__ a, float4* __restrict__ b, size_t n4){ size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n4;i+=st) b[i]=a[i]; }
template<int MODE,int CH> float run(const float* src,int n,Aff A,float* dst,int reps){
  dim3 grid((n+64*CH-1)/(64*CH),(n+3)/4,n), block(256);
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int i=0;i<3;++i) hipLaunchKernelGGL((kp<MODE,CH>),grid,block,0,0,src,n,n,n,A,dst,n,n,n);
  CK(hipEventRecord(e0));
  for(int i=0;i<reps;++i) hipLaunchKernelGGL((kp<MODE,CH>),grid,block,0,0,src,n,n,n,A,dst,n,n,n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms*1e3f/reps;
}
Aff rigid(float tx,float ty,float tz,float rx,float ry,float rz){
  float cx=cosf(rx),sx=sinf(rx),cy=cosf(ry),sy=sinf(ry),cz=cosf(rz),sz=sinf(rz);
  float Rx[9]={1,0,0,0,cx,-sx,0,sx,cx},Ry[9]={cy,0,sy,0,1,0,-sy,0,cy},Rz[9]={cz,-sz,0,sz,cz,0,0,0,1};
  float T[9],R[9];
  for(int i=0;i<3;++i)for(int j=0;j<3;++j){T[i*3+j]=0;for(int k=0;k<3;++k)T[i*3+j]+=Ry[i*3+k]*Rx[k*3+j];}
  for(int i=0;i<3;++i)for(int j=0;j<3;++j){R[i*3+j]=0;for(int k=0;k<3;++k)R[i*3+j]+=Rz[i*3+k]*T[k*3+j];}
  Aff A; for(int i=0;i<3;++i){for(int j=0;j<3;++j)A.m[i*4+j]=R[i*3+j];} A.m[3]=tx;A.m[7]=ty;A.m[11]=tz; return A;
}
int main(){
  const int n=256; size_t N=(size_t)n*n*n;
  float *src,*dst; CK(hipMalloc(&src,N*4)); CK(hipMalloc(&dst,N*4));
  std::vector<float> h(N); for(size_t i=0;i<N;++i) h[i]=(float)(i%977)*0.001f; CK(hipMemcpy(src,h.data(),N*4,hipMemcpyHostToDevice));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int i=0;i<3;++i) hipLaunchKernelGGL(kcopy,dim3(4096),dim3(256),0,0,(const float4*)src,(float4*)dst,N/4);
  CK(hipEventRecord(e0)); for(int i=0;i<20;++i) hipLaunchKernelGGL(kcopy,dim3(4096),dim3(256),0,0,(const float4*)src,(float4*)dst,N/4);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  printf("copy 67MB->67MB: %.1f us (%.2f TB/s)\n", ms*1e3/20, 2*N*4/(ms*1e-3/20)/1e12);
  Aff ids[3]={rigid(2.3f,-1.7f,3.1f,0,0,0), rigid(2.3f,-1.7f,3.1f,0.05f,-0.08f,0.03f), rigid(2.3f,-1.7f,3.1f,0.1f,0.1f,0.1f)};
  const char* nm[3]={"rot0","rot.05","rot.1"};
  for(int a=0;a<3;++a){
    printf("%s: x2-lerp CH4 %.1f | CH1 %.1f | CH2 %.1f | noload %.1f | 8xdword %.1f | 1load %.1f | 4dword %.1f | 2x(x2) %.1f us\n", nm[a],
      run<0,4>(src,n,ids[a],dst,20), run<0,1>(src,n,ids[a],dst,20), run<0,2>(src,n,ids[a],dst,20), run<1,4>(src,n,ids[a],dst,20), run<2,4>(src,n,ids[a],dst,20), run<3,4>(src,n,ids[a],dst,20), run<4,4>(src,n,ids[a],dst,20), run<5,4>(src,n,ids[a],dst,20));
  }
  return 0;
}
