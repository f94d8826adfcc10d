/* oracle_collide.c — collision stage of the CPU restatement (A5/A6). TEST INFRASTRUCTURE ONLY.
 * Placeholder until the narrowphase lands: contacts disabled => ncon = 0. */
#include "oracle.h"

void orc_collision(const orc_model *m, orc_data *d) {
  (void)m;
  d->ncon = 0;
}
int orc_narrow_pair(const orc_model *m, const orc_data *d, int g1, int g2, double margin,
                    double *dist, double *pos, double *normal) {
  (void)m; (void)d; (void)g1; (void)g2; (void)margin; (void)dist; (void)pos; (void)normal;
  return 0;
}
