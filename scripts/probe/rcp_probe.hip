// accuracy of v_rcp_f64 + n Newton steps against IEEE division, inputs in [0.3, 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* a, double* r0, double* r1, double* r2, int n){
  int i = blockIdx.x*blockDim.x+threadIdx.x; if(i>=n) return;
  double x=a[i]; double r=__builtin_amdgcn_rcp(x); r0[i]=r;
  double e=__builtin_fma(-x,r,1.0); r=__builtin_fma(r,e,r); r1[i]=r;
  e=__builtin_fma(-x,r,1.0); r=__builtin_fma(r,e,r); r2[i]=r; }
int main(){ int n=1<<20; std::vector<double> h(n); for(int i=0;i<n;i++) h[i]=0.3+7.7*((i*2654435761u)%1000003)/1000003.0;
  double *a,*r0,*r1,*r2; hipMalloc(&a,n*8);hipMalloc(&r0,n*8);hipMalloc(&r1,n*8);hipMalloc(&r2,n*8);
  hipMemcpy(a,h.data(),n*8,hipMemcpyHostToDevice); k<<<n/256,256>>>(a,r0,r1,r2,n);
  std::vector<double> o0(n),o1(n),o2(n); hipMemcpy(o0.data(),r0,n*8,hipMemcpyDeviceToHost);hipMemcpy(o1.data(),r1,n*8,hipMemcpyDeviceToHost);hipMemcpy(o2.data(),r2,n*8,hipMemcpyDeviceToHost);
  double m0=0,m1=0,m2=0; for(int i=0;i<n;i++){ double t=1.0/h[i]; m0=fmax(m0,fabs(o0[i]-t)/t); m1=fmax(m1,fabs(o1[i]-t)/t); m2=fmax(m2,fabs(o2[i]-t)/t);} 
  printf("max rel err: rcp %.3e  +1NR %.3e  +2NR %.3e  (eps=%.3e)\n",m0,m1,m2,2.22e-16); return 0; }
