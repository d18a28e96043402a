/*
 * oracle/arx_oracle_rfa.c -- TEST INFRASTRUCTURE ONLY.
 * CPU restatement of the Go half of the path (candidate post-processing and the RFA scorer).
 * Filled in by a later milestone; kept as a separate translation unit.
 */
#include "arx_oracle.h"
