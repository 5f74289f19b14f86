/* fake_roctx.c — TEST INFRASTRUCTURE: a stand-in for the roctx library (YART_ROCTX_LIB names it; csrc/trace_ranges.hpp binds
 * roctxRangePushA / roctxRangePop / roctxMarkA by name). It appends one line per call to the file FAKE_ROCTX_LOG names, so that a
 * test can check which ranges a render opens, that they nest, and that every push has its pop. */
#include <stdio.h>
#include <stdlib.h>

static int depth = 0;
static void line(const char* what, const char* name) {
  const char* path = getenv("FAKE_ROCTX_LOG");
  if (!path || !*path) return;
  FILE* f = fopen(path, "a");
  if (!f) return;
  fprintf(f, "%s %d %s\n", what, depth, name ? name : "");
  fclose(f);
}
int roctxRangePushA(const char* name) { line("push", name); return depth++; }
int roctxRangePop(void) { depth--; line("pop", ""); return depth; }
void roctxMarkA(const char* name) { line("mark", name); }
