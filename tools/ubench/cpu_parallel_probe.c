/* cpu_parallel_probe.c -- how many CPUs does this pod really get?  Times the same arithmetic on 1, 2, 4 .. 256 OpenMP threads: the
 * speed-up saturates at the cgroup CPU quota (round-6 GPU boxes: 16 of the host's 256 hardware threads -- `cat /sys/fs/cgroup/cpu.max`),
 * and more busy threads than that are throttled.  gcc -O1 -fopenmp tools/ubench/cpu_parallel_probe.c -o tools/ubench/cpu_parallel_probe */
#include <omp.h>
#include <stdio.h>
#include <time.h>
static double now(){struct timespec t; clock_gettime(CLOCK_MONOTONIC,&t); return t.tv_sec+t.tv_nsec*1e-9;}
int main(){ for (int nt=1; nt<=256; nt*=2){ double t0=now(); double tot=0; 
#pragma omp parallel num_threads(nt) reduction(+:tot)
 { double a=0; for(long i=0;i<400000000L;++i) a+=i*1e-9; tot+=a; }
 double dt=now()-t0; printf("%3d threads: %.3f s  (%.1f x one thread's work per second) %g\n", nt, dt, nt/dt*0.0+ (double)nt/dt, tot>0?0.0:1.0);} return 0; }
