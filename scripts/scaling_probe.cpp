// Probe (not product): does the host side of a GPU box scale with threads?  g++ -O2 -fopenmp scripts/scaling_probe.cpp -o scripts/_scaling
#include <omp.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
#include <sched.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <initializer_list>
static double now(){return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(){
  { FILE*f=fopen("/sys/fs/cgroup/cpu.max","r"); char b[128]; if(f){ if(fgets(b,128,f)) printf("cpu.max: %s",b); fclose(f);} 
    cpu_set_t s; sched_getaffinity(0,sizeof(s),&s); printf("affinity CPUs: %d\n",CPU_COUNT(&s)); }
  const size_t N=10u<<20; std::vector<unsigned> key(N); for(size_t i=0;i<N;i++) key[i]=(unsigned)i;
  // a shared file of 1.6 GB
  const size_t FB=1600u<<20; { int fd=open("/tmp/_probe_file",O_WRONLY|O_CREAT|O_TRUNC,0644); std::vector<char> blk(8<<20,'7'); for(size_t i=0;i<blk.size();i+=37) blk[i]='\n'; for(size_t w=0;w<FB;w+=blk.size()) if(write(fd,blk.data(),blk.size())<0) return 1; close(fd);} 
  for(int th: {1,4,16}){ omp_set_num_threads(th); double t0; 
    t0=now(); double s=0;
#pragma omp parallel reduction(+:s)
    { unsigned long x=omp_get_thread_num()+1; for(long i=0;i<800000000L/omp_get_num_threads();i++){ x=x*6364136223846793005UL+1442695040888963407UL; } s+=x; }
    printf("%2d threads: ALU %.3f", th, now()-t0);
    t0=now(); unsigned long acc=0;
#pragma omp parallel reduction(+:acc)
    { unsigned long x=omp_get_thread_num()+1; for(long i=0;i<64000000L/omp_get_num_threads();i++){ x=x*6364136223846793005UL+1442695040888963407UL; acc+=key[(x>>33)%N]; } }
    printf("  random reads %.3f", now()-t0);
    t0=now(); { int fd=open("/tmp/_probe_file",O_RDONLY); const char*m=(const char*)mmap(0,FB,PROT_READ,MAP_PRIVATE,fd,0); close(fd); unsigned long lines=0;
#pragma omp parallel reduction(+:lines)
      { size_t lo=FB*omp_get_thread_num()/omp_get_num_threads(), hi=FB*(omp_get_thread_num()+1)/omp_get_num_threads(); for(const char*q=m+lo,*e=m+hi;q<e;){ const char*nl=(const char*)memchr(q,'\n',e-q); if(!nl) break; lines++; q=nl+1; } }
      munmap((void*)m,FB); printf("  mmap file scan %.3f (%lu)", now()-t0, lines); }
    for(int huge=0;huge<2;huge++){ t0=now(); const size_t B=1600u<<20; char*p=(char*)mmap(0,B,PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS|MAP_NORESERVE,-1,0); if(huge) madvise(p,B,MADV_HUGEPAGE);
#pragma omp parallel
      { size_t lo=B*omp_get_thread_num()/omp_get_num_threads(), hi=B*(omp_get_thread_num()+1)/omp_get_num_threads(); for(size_t i=lo;i<hi;i+=64) p[i]=1; }
      double t1=now(); munmap(p,B); printf("  first touch %s %.3f (+unmap %.3f)", huge?"2M":"4K", t1-t0, now()-t1); }
    { struct C{ long a; }; std::vector<C> c(64); t0=now();
#pragma omp parallel
      { volatile long*q=&c[omp_get_thread_num()].a; for(long i=0;i<200000000L/omp_get_num_threads();i++) (*q)++; }
      printf("  false sharing %.3f", now()-t0); }
    printf("\n"); }
  unlink("/tmp/_probe_file");
}
