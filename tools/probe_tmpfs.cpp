#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <thread>
#include <vector>
#include <chrono>
static double now(){return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(int argc,char**argv){
  size_t mb = argc>1?atol(argv[1]):2048; int T = argc>2?atoi(argv[2]):16; size_t n = mb<<20;
  char *src=(char*)malloc(n); memset(src,'x',n);
  const char*path="/dev/shm/xm_probe_tmp";
  for (int mode=0; mode<7; ++mode){
    unlink(path); int fd=open(path,O_RDWR|O_CREAT,0600);
    double t0=now();
    if(mode==0||mode==5||mode==6){ posix_fallocate(fd,0,n); }
    else ftruncate(fd,n);
    char*m=(char*)mmap(0,n,PROT_READ|PROT_WRITE,MAP_SHARED|(mode==6?MAP_POPULATE:0),fd,0);
    double t1=now();
    std::vector<std::thread> th;
    size_t per=n/T;
    for(int t=0;t<T;++t) th.emplace_back([&,t]{
      size_t b=t*per,e=(t==T-1)?n:b+per;
      if(mode==0||mode==1||mode==6){ memcpy(m+b,src+b,e-b);}
      else if(mode==5){ madvise(m+b,e-b,23); memcpy(m+b,src+b,e-b);}            // 0: after fallocate; 1: demand faults
      else if(mode==2){ madvise(m+b,e-b,23 /*MADV_POPULATE_WRITE*/); memcpy(m+b,src+b,e-b);}
      else if(mode==3){ size_t o=b; while(o<e){ ssize_t g=pwrite(fd,src+o,std::min<size_t>(e-o,8<<20),o); if(g<=0)break; o+=g;} }
      else if(mode==4){ // fallocate per thread range
         posix_fallocate(fd,b,e-b); memcpy(m+b,src+b,e-b);}
    });
    for(auto&x:th)x.join();
    double t2=now();
    const char*names[]={"fallocate+copy","faults+copy","populate_write+copy","pwrite parallel","fallocate per thread+copy","fallocate+populate_write+copy","fallocate+MAP_POPULATE+copy"};
    printf("%-28s prep %.3f s  fill %.3f s  total %.2f GB/s\n",names[mode],t1-t0,t2-t1,n/1e9/(t2-t0));
    double t3=now(); munmap(m,n); printf("      munmap %.3f s\n", now()-t3); close(fd); unlink(path);
  }
}
