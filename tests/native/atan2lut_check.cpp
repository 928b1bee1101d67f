// Host-side check of mx::atan2lut (mods_amd/csrc/kmath.hpp, branch-free) against the eight-way branch form of
// atan2LUTff (reference: detectors/helpers.cpp:160-207), restated below: 20 M random and special-case arguments
// (equal magnitudes, zeros of both signs, axis-aligned, huge / tiny) must agree bit for bit, signs of zero included.
// Built and run by tests/test_host_abi.py::test_atan2lut_branchfree with hipcc (kmath.hpp needs the HIP headers).
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdint>
#include "kmath.hpp"
static float ref(const double*L,float y,float x){
  const float PI_2f = 1.57079632679489661923f, PIf = 3.14159265358979323846f;
  if (x > 0.f) {
    if (y > 0.f) {
      if (x > y) return (float)L[(int)(255.f * y / x)];
      return (float)((double)PI_2f - L[(int)(255.f * x / y)]);
    }
    float ay = fabsf(y);
    if (x > ay) return (float)(-L[(int)(255.f * ay / x)]);
    return (float)((double)(-PI_2f) + L[(int)(255.f * x / ay)]);
  }
  if (y > 0.f) {
    float ax = fabsf(x);
    if (ax > y) return (float)((double)PIf - L[(int)(255.f * y / ax)]);
    return (float)((double)PI_2f + L[(int)(255.f * ax / y)]);
  }
  float ax = fabsf(x), ay = fabsf(y);
  if (ax > ay) return (float)((double)(-PIf) + L[(int)(255.f * ay / ax)]);
  if (x == 0.f) return 0.f;
  return (float)((double)(-PI_2f) - L[(int)(255.f * ax / ay)]);
}
int main(){ double L[256]; for(int i=0;i<256;i++) L[i]=atan(i/255.0);
  unsigned long long bad=0,n=0; uint32_t st=12345;
  auto rnd=[&](){st=st*1664525u+1013904223u; return st;};
  float specials[]={0.f,-0.f,1.f,-1.f,1e-30f,-1e-30f,255.f,-255.f,3.5f,-3.5f,1e20f,-1e20f};
  for(float x:specials)for(float y:specials){float a=ref(L,y,x),b=mx::atan2lut(L,y,x);n++; if(memcmp(&a,&b,4)){bad++; printf("bad %g %g: %g %g\n",y,x,a,b);} }
  for(long i=0;i<20000000;i++){ float x=((int)(rnd()>>8)-(1<<23))/65536.f, y=((int)(rnd()>>8)-(1<<23))/65536.f; if(i%7==0) y=x; if(i%11==0) y=-x; if(i%13==0) x=0; if(i%17==0)y=0;
    float a=ref(L,y,x),b=mx::atan2lut(L,y,x);n++; if(memcmp(&a,&b,4)){bad++; if(bad<10)printf("bad %g %g: %g %g\n",y,x,a,b);} }
  printf("n=%llu bad=%llu\n",n,bad); return bad!=0; }
