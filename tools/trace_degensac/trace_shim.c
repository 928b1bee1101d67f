/* TEST INFRASTRUCTURE: print-and-forward wrappers for the call sites of the reference's exp_ranF.c (tools/trace_degensac/build.sh renames
 * them with -Dname=tr_name at compile time; no reference source is modified or copied). */
#include <stdio.h>
typedef struct { unsigned I; double J; } Score;
int checksample(double * F, double * u7, double th, double * H);
unsigned rFtH(double * u, unsigned char * hinl, double th, double * H, unsigned len, double *F, int *p, double *b);
unsigned innerH(double * H, double * u, unsigned len, double th, unsigned iters, unsigned char * inl, int * pool, double * buffer);
int nsamples(int ninl, int ptNum, int samsiz, double conf);
Score inlidxs (const double * err, int len, double th, int * inl);
int nullspace(double *matrix, double *nullspace, int n, int * buffer);
int rroots3 (double *po, double *r);
int all_ori_valid(double *F, double *us, int *idx, int N);
int tr_checksample(double * F, double * u7, double th, double * H) { int r = checksample(F,u7,th,H); fprintf(stderr,"cs %d H0 %.17g\n", r, H[0]); return r; }
unsigned tr_rFtH(double * u, unsigned char * hinl, double th, double * H, unsigned len, double *F, int *p, double *b) { unsigned r = rFtH(u,hinl,th,H,len,F,p,b); fprintf(stderr,"rFtH %u F0 %.17g\n", r, F[0]); return r; }
unsigned tr_innerH(double * H, double * u, unsigned len, double th, unsigned iters, unsigned char * inl, int * pool, double * buffer) { unsigned r = innerH(H,u,len,th,iters,inl,pool,buffer); fprintf(stderr,"innerH %u H0 %.17g\n", r, H[0]); return r; }
int tr_nsamples(int ninl, int ptNum, int samsiz, double conf) { int r = nsamples(ninl,ptNum,samsiz,conf); fprintf(stderr,"nsamples %d %d -> %d\n", ninl, ptNum, r); return r; }
Score tr_inlidxs (const double * err, int len, double th, int * inl) { Score s = inlidxs(err,len,th,inl); fprintf(stderr,"S %u %.17g th %.3g\n", s.I, s.J, th); return s; }
int tr_nullspace(double *m, double *ns, int n, int * buffer) { int r = nullspace(m,ns,n,buffer); fprintf(stderr,"null %d\n", r); return r; }
int tr_rroots3 (double *po, double *r) { int n = rroots3(po,r); fprintf(stderr,"roots %d %.17g\n", n, n?r[0]:0.0); return n; }
int tr_all_ori_valid(double *F, double *us, int *idx, int N) { int r = all_ori_valid(F,us,idx,N); fprintf(stderr,"ori %d\n", r); return r; }
