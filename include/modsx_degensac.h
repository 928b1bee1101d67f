/*
 * modsx_degensac.h -- the reference's verification entry points, exported by libmodsx.so with the reference's own names
 * and signatures (degensac/exp_ranH.h:29-36, degensac/exp_ranF.h:67-72, degensac/Htools.h:1-3, degensac/Fcustomdef.h:3-4,
 * degensac/rtools.h:17-24), so that LORANSACFiltering (matching/matching.cpp:806-980) links against libmodsx unchanged.
 * See mods_amd/csrc/ransac_shim.cpp for what is and is not reproduced (resids, H_best, Ih, foreign error functions).
 */
#ifndef MODSX_DEGENSAC_H
#define MODSX_DEGENSAC_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { unsigned I; double J; } Score;
typedef void (*HDsPtr)(const double *, const double *, const double *, double *, int);
typedef void (*HDsiPtr)(const double *, const double *, const double *, double *, int, int *, int);
typedef void (*HDsidxPtr)(const double *, const double *, const double *, double *, int, int *, int);
typedef void (*FDsPtr)(const double *, const double *, double *, int);
typedef void (*exFDsPtr)(const double *, const double *, double *, double *, int);

/* Sampson error of a homography (Htools.c:158-196, 284-320, 418-456); `lin` as lin_hg() lays it out */
void HDs(const double *lin, const double *u, const double *H, double *p, int len);
void HDsi(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni);
void HDsidx(const double *lin, const double *u6, const double *H, double *p, int len, int *idx, int siz);
/* symmetric transfer error of a homography: sum (SYMM_SUM) and max (SYMM_MAX) of the two directions
 * (Htools.c:199-279, 325-411, 458-535) -- the triples matching.cpp:834-846 takes the addresses of */
void HDsSym(const double *lin, const double *u, const double *H, double *p, int len);
void HDsSymMax(const double *lin, const double *u, const double *H, double *p, int len);
void HDsiSym(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni);
void HDsiSymMax(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni);
void HDsSymidx(const double *lin, const double *mu, const double *H, double *p, int len, int *idx, int siz);
void HDsSymidxMax(const double *lin, const double *mu, const double *H, double *p, int len, int *idx, int siz);
/* Sampson / symmetric epipolar error of a fundamental matrix (Ftools.c:82-210) */
void FDs(const double *u, const double *F, double *p, int len);
void FDsSym(const double *u, const double *F, double *p, int len);
void exFDs(const double *u, const double *F, double *p, double *w, int len);
void exFDsSym(const double *u, const double *F, double *p, double *w, int len);

Score exp_ransacHcustom(double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl, int iter_type,
                        int *data_out, int oriented_constraint, unsigned inlLimit, double **resids, HDsPtr HDS1,
                        HDsiPtr HDSi1, HDsidxPtr HDSidx1, int doSymCheck);
int exp_ransacFcustom(double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl, int *data_out,
                      int do_lo, unsigned inlLimit, double **resids, double *H_best, int *Ih, exFDsPtr EXFDS1, FDsPtr FDS1,
                      int doSymCheck);

/* The reference seeds with srand(time(NULL)) (exp_ranH.c:823, exp_ranF.c:822).  enable != 0 fixes the seed of every later
 * call (MODSX_RANSAC_SEED in the environment does the same); enable == 0 returns to time seeding. */
void modsx_ransac_set_seed(unsigned seed, int enable);

#ifdef __cplusplus
}
#endif
#endif
