// Host-side fuzz harness for the FLAC decoder (csrc/flac.cpp) under AddressSanitizer: a valid stream is damaged
// at random (byte flips, truncation, deleted and inserted runs, header damage), copied into an exact-size heap block
// and decoded with 1, 2 and 5 threads, to int32 and to int16, whole and by random sample ranges.  Any read or write
// outside a buffer aborts the run.
// Build and run (host code only; no GPU is touched):
//   hipcc -O1 -g -std=c++17 -Xarch_host -fsanitize=address -Iinclude -Iomnilingual-asr_amd/csrc -x hip \
//         --offload-arch=gfx950 -c omnilingual-asr_amd/csrc/flac.cpp -o /tmp/flac_asan.o
//   g++ -O1 -g -fsanitize=address -no-pie tools/flac_fuzz.cpp /tmp/flac_asan.o -o /tmp/flac_fuzz \
//         -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread
//   ASAN_OPTIONS=detect_leaks=0 /tmp/flac_fuzz tests/golden/reference_upload.flac 1500
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <vector>
#include <random>
#include <algorithm>
namespace oasr { void set_error(const char*, ...) {} }   // the library's error text is not needed here
extern "C" const char* oasr_last_error() { return ""; }
extern "C" int oasr_flac_decode(const uint8_t*, int64_t, int32_t*, int64_t, int64_t*);
extern "C" int oasr_flac_decode16(const uint8_t*, int64_t, int16_t*, int64_t, int64_t*);
extern "C" int oasr_flac_info(const uint8_t*, int64_t, int32_t*, int32_t*, int32_t*, int64_t*, uint8_t*);
extern "C" int oasr_flac_decode_range(const uint8_t*, int64_t, int64_t, int64_t, void*, int32_t);
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET);
  std::vector<uint8_t> d(n); if(fread(d.data(),1,n,f)!=(size_t)n) return 2;
  int iters=atoi(argv[2]);
  std::mt19937_64 rng(7);
  int32_t sr,ch,bits; int64_t total; 
  oasr_flac_info(d.data(),n,&sr,&ch,&bits,&total,nullptr);
  int ok=0,err=0;
  for(int it=0;it<iters;it++){
    std::vector<uint8_t> b(d);
    int kind=rng()%5;
    if(kind==0){ int k=1+rng()%6; for(int i=0;i<k;i++) b[rng()%b.size()]^=(uint8_t)(1+rng()%255); }
    else if(kind==1){ b.resize(1+rng()%b.size()); }
    else if(kind==2){ size_t i=rng()%(b.size()-10); size_t m=1+rng()%4000; if(i+m>b.size()) m=b.size()-i; b.erase(b.begin()+i,b.begin()+i+m); }
    else if(kind==3){ size_t i=rng()%b.size(); size_t m=1+rng()%3000; std::vector<uint8_t> ins(m); for(auto&x:ins) x=(uint8_t)rng(); b.insert(b.begin()+i,ins.begin(),ins.end()); }
    else { for(int i=4;i<60&&i<(int)b.size();i++) if(rng()%4==0) b[i]=(uint8_t)rng(); }   // header damage
    // exact-size heap copy so that ASan sees any read past the end
    uint8_t* p=(uint8_t*)malloc(b.size()); memcpy(p,b.data(),b.size());
    int32_t s2,c2,b2; int64_t t2=0;
    int rc=oasr_flac_info(p,b.size(),&s2,&c2,&b2,&t2,nullptr);
    if(rc==0 && t2>0 && t2<(1<<26) && c2>0){
      const char* th[]={"1","2","5"}; setenv("OASR_FLAC_THREADS",th[rng()%3],1);
      std::vector<int32_t> out((size_t)t2*c2); int64_t w=0;
      int r1=oasr_flac_decode(p,b.size(),out.data(),t2,&w);
      if(b2<=16){ std::vector<int16_t> o16((size_t)t2*c2); oasr_flac_decode16(p,b.size(),o16.data(),t2,&w); }
      // sample ranges (what a rank of a process group decodes): exact-size outputs, both forms
      for(int q=0;q<3;q++){
        int64_t first=(int64_t)(rng()%(uint64_t)t2), count=1+(int64_t)(rng()%(uint64_t)(t2-first));
        if(rng()%4==0) count=std::min<int64_t>(count,1+(int64_t)(rng()%5000));
        if(b2<=16 && (rng()&1)){ int16_t* o=(int16_t*)malloc((size_t)count*c2*2); oasr_flac_decode_range(p,b.size(),first,count,o,16); free(o); }
        else { int32_t* o=(int32_t*)malloc((size_t)count*c2*4); oasr_flac_decode_range(p,b.size(),first,count,o,32); free(o); }
      }
      if(r1==0) ok++; else err++;
    } else err++;
    free(p);
  }
  printf("asan fuzz: ok %d err %d\n",ok,err);
}
