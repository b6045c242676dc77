#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
int main(){
  long bad1=0,bad2=0,tot=0;
  for(int k=0;k<=22;k++){
    double T=1; for(int i=0;i<k;i++) T*=10.0;   /* exact up to 1e22 */
    double Ti=1.0/T;
    for(long N=1000;N<=1000000;N++){
      if(N>10000 && N<100000) continue;
      double n=(double)N;
      double want=n/T;
      double q=n*Ti; double r=fma(-q,T,n); double q1=fma(r,Ti,q);
      double r1=fma(-q1,T,n); double q2=fma(r1,Ti,q1);
      tot++;
      if(q1!=want) bad1++;
      if(q2!=want) bad2++;
    }
  }
  printf("tot %ld bad after 1 step %ld, after 2 steps %ld\n",tot,bad1,bad2);
  /* random general division: a/d with y=RN(1/d): 2-step */
  srand48(1); long badg=0, badg1=0; long n=200000000;
  for(long i=0;i<n;i++){
    double a=(drand48()-0.5)*pow(2.0,(int)(drand48()*40)-20);
    double d=(drand48()+0.01)*pow(2.0,(int)(drand48()*20)-10);
    double y=1.0/d; double want=a/d;
    double q=a*y; double r=fma(-q,d,a); double q1=fma(r,y,q); double r1=fma(-q1,d,a); double q2=fma(r1,y,q1);
    if(q2!=want) badg++;
    if(q1!=want) badg1++;
  }
  printf("general: %ld cases, 1-step mismatches %ld, 2-step mismatches %ld\n",n,badg1,badg);
  return 0;
}
