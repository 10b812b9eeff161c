#include "../../include/ngm_pipeline.h"
extern "C" {
int ngm_ref_decode(const ngm_ref *, uint64_t, int, char *) { return -38; }
ngm_mapper *ngm_mapper_create(const ngm_ref *, const ngm_mapper_params *) { return 0; }
void ngm_mapper_destroy(ngm_mapper *) {}
int ngm_mapper_cs(ngm_mapper *, int, const char *, uint32_t *, float *) { return -38; }
int ngm_mapper_cs_fetch(ngm_mapper *, uint64_t *, uint8_t *, float *) { return -38; }
int ngm_mapper_map_se(ngm_mapper *, int, const char *, ngm_hit *, char *, char *) { return -38; }
int ngm_mapper_last_kernel_ms(ngm_mapper *, float *) { return -38; }
}
