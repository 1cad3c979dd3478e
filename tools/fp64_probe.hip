// probe: which fp64 primitives are correctly rounded on gfx950 under the flags we build with
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(int side, double* s, double* q, double* s2, double *rn) {
    int t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= side*side) return;
    int dx = t / side, dy = t % side; long long d2 = (long long)dx*dx + (long long)dy*dy;
    double x = (double)d2;
    double r = sqrt(x); s[t] = r; q[t] = (r == 0.0) ? 1.0 : 40.0 / r;
    s2[t] = __dsqrt_rn(x);
    // Markstein-style correction
    double g = r; if (x > 0) { double h = 0.5 / g; double e = fma(-g, g, x); g = fma(h, e, g); }
    rn[t] = g;
}
int main() {
    int side = 513; size_t n = (size_t)side*side;
    double *s,*q,*s2,*rn; hipMalloc(&s,n*8); hipMalloc(&q,n*8); hipMalloc(&s2,n*8); hipMalloc(&rn,n*8);
    k<<<(n+255)/256,256>>>(side,s,q,s2,rn); 
    std::vector<double> hs(n),hq(n),hs2(n),hrn(n);
    hipMemcpy(hs.data(),s,n*8,hipMemcpyDeviceToHost); hipMemcpy(hq.data(),q,n*8,hipMemcpyDeviceToHost);
    hipMemcpy(hs2.data(),s2,n*8,hipMemcpyDeviceToHost); hipMemcpy(hrn.data(),rn,n*8,hipMemcpyDeviceToHost);
    long bad_s=0,bad_q=0,bad_s2=0,bad_rn=0,bad_hyp=0, bad_q_given=0;
    for (int dx=0;dx<side;dx++) for(int dy=0;dy<side;dy++){ size_t t=(size_t)dx*side+dy;
        double x=(double)((long long)dx*dx+(long long)dy*dy); double r=std::sqrt(x); double h=std::hypot((double)dx,(double)dy);
        if (r!=h) bad_hyp++;
        if (hs[t]!=r) bad_s++; if (hs2[t]!=r) bad_s2++; if (hrn[t]!=r) bad_rn++;
        double qq = (r==0.0)?1.0:40.0/r; if (hq[t]!=qq) bad_q++;
        double qg = (hs[t]==0.0)?1.0:40.0/hs[t]; if (hq[t]!=qg) bad_q_given++;
    }
    printf("host hypot!=sqrt %ld | dev sqrt bad %ld | __dsqrt_rn bad %ld | corrected bad %ld | div bad (total) %ld | div bad given same operand %ld\n", bad_hyp,bad_s,bad_s2,bad_rn,bad_q,bad_q_given);
    return 0;
}
