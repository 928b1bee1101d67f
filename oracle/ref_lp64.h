/* oracle/ref_lp64.h -- TEST INFRASTRUCTURE ONLY: force-included (gcc -include) when oracle/Makefile compiles the
 * reference's LAPACK-binding sources (degensac/lapwrap.c and the files that include lapwrap.h) in place.
 *
 * degensac/lapwrap.h:12 declares `typedef ptrdiff_t lapack_int;` and lapwrap.c passes `&info` of that type to
 * dgesvd_/dsyev_.  LAPACK's INTEGER is 32 bits (LP64 interface, also in the OpenBLAS this build links), so on a
 * 64-bit host only the low half of `info` is written and `if (info != 0)` (lapwrap.c:45,90) tests four
 * uninitialised stack bytes: singulF (Ftools.c:292-312) then replaces F by the identity whenever that garbage is
 * non-zero.  The code is well defined where ptrdiff_t is the LAPACK integer (the 32-bit targets it was written
 * for); this prelude reproduces exactly that configuration.  No reference source is modified or copied. */
#include <stddef.h>
#define ptrdiff_t int
