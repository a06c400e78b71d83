#include <stdint.h>
int hnsw_index_info(const void *a, int64_t *n, int32_t *d, int32_t *m, int32_t *mm) { return 1; }
int hnsw_index_graph_size(const void *a, int64_t *b, int64_t *c, int64_t *d, int32_t *e) { return 1; }
int hnsw_index_graph(const void *a, int32_t *b, int64_t *c, int64_t *d, int64_t *e) { return 1; }
int hnsw_index_get_ids(const void *a, int64_t *b) { return 1; }
const char *hnsw_last_error(void) { return "stub"; }
int hnsw_index_build() { return 1; }
