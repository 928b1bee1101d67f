/*
 * ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 * Determinism injection for the reference's degensac build (oracle/_ref):
 * exp_ranH.c / exp_ranF.c seed libc's PRNG with srand(time(NULL))
 * (degensac/exp_ranH.c:823, exp_ranF.c:822).  Those two translation units are
 * compiled with -Dtime=modsx_ref_time so the seed comes from here instead of the
 * wall clock; no reference file is modified or copied.
 */
static unsigned g_seed = 1;
void modsx_ref_set_seed(unsigned s) { g_seed = s; }
long modsx_ref_time(long *t) { if (t) *t = (long)g_seed; return (long)g_seed; }
