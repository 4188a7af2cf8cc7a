// sp_count2.hip -- K1 engine 2: LDS radix-partition counter (placeholder until measured design lands).
#include "sp_device.h"

int sp_count_engine2(sp_ctx *ctx, sp_chrom &c, const sp_kparams &kp, int lower, unsigned long long *d_len2) {
    (void)c; (void)kp; (void)lower; (void)d_len2;
    return sp_fail(ctx, SP_EUNSUP, "count engine 2 is not built yet");
}
