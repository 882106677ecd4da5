/*
 * thompson_oracle_p32n.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * The reference's NATIVE arithmetic ("P32n" of SURVEY.md 8c: REAL = binary32 state and work variables,
 * DOUBLE PRECISION = binary64 process rates, as the Fortran is shipped).  This file only re-compiles
 * thompson_oracle_column.c under -DTH_P32N; the Makefile adds -fsingle-precision-constant so that unsuffixed
 * literals are binary32 like Fortran's default-REAL literals.  See the header of thompson_oracle_column.c.
 *
 * Not P32n (documented in DESIGN.md): the lookup tables are the P64 build's (the reference's tables are
 * binary64 in both builds, but its builders mix REAL constants into them: ~1e-7 relative differences).
 *
 * Pinned on the native known answers the survey recorded: tests/test_oracle_p32n.py.
 */
#define TH_P32N 1
#include "thompson_oracle_column.c"
