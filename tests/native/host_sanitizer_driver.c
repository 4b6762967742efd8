#include "dlpm_amd.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(void) {
    for (int T = 2; T <= 1000; T = T * 3 + 1) {
        float *g = malloc(4 * T), *bg = malloc(4 * T), *s = malloc(4 * T), *bs = malloc(4 * T);
        if (dlpm_schedule_f32(T, 1.7, g, bg, s, bs) != DLPM_OK) return 2;
        if (dlpm_schedule_exploding_f32(T, 1.8, g, bg, s, bs) != DLPM_OK) return 3;
        free(g); free(bg); free(s); free(bs);
    }
    dlpm_mt19937 st;
    dlpm_mt19937_seed(&st, 5u);
    for (int n = 1; n <= 70; n += 3) {
        float *o = malloc(4 * n);
        if (dlpm_randn_host_f32(&st, n, o) != DLPM_OK) return 4;
        if (dlpm_skewed_levy_host_f32(&st, 1.7, n, n % 2 ? 10.0 : -1.0, o) != DLPM_OK) return 5;
        free(o);
    }
    for (int steps = 1; steps <= 40; steps += 13) {
        float *b = malloc(4 * 5 * (steps + 1));
        if (dlpm_lim_tables_f32(1.8, steps, steps & 1, b, b + (steps + 1), b + 2 * (steps + 1), b + 3 * (steps + 1), b + 4 * (steps + 1)) != DLPM_OK) return 6;
        free(b);
    }
    for (int H = 1; H <= 33; H += 8) {
        int W = H + 3;
        unsigned char *img = malloc(3 * H * W);
        for (int i = 0; i < 3 * H * W; i++) img[i] = (unsigned char)(i * 7);
        long long cap = dlpm_png_bound(H, W), used = 0;
        unsigned char *out = malloc(cap);
        if (dlpm_png_encode_rgb8(img, H, W, 6, out, cap, (int64_t *)&used) != DLPM_OK || used <= 0 || used > cap) return 7;
        free(out); free(img);
    }
    float bad[4];
    if (dlpm_schedule_f32(4, 2.5, bad, bad, bad, bad) == DLPM_OK) return 8;   /* error channel */
    printf("ok %s\n", dlpm_last_error());
    return 0;
}
