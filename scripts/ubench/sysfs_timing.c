/* sysfs_timing.c -- what does it cost to find a GPU's NUMA node without the HIP runtime?  (measurement aid) */
#define _GNU_SOURCE
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void cat(const char *path, int show)
{
    char buf[8192];
    double t0 = now();
    FILE *f = fopen(path, "r");
    size_t n = f ? fread(buf, 1, sizeof buf - 1, f) : 0;
    if (f) fclose(f);
    buf[n] = 0;
    printf("%8.3f ms  %s  (%zu bytes)\n", 1e3 * (now() - t0), path, n);
    if (show && n) { for (char *q = buf; *q; ++q) if (*q == '\n') *q = ' '; printf("            %.400s\n", buf); }
}
int main(void)
{
    char p[256];
    for (int i = 0; i < 12; ++i) {
        snprintf(p, sizeof p, "/sys/class/kfd/kfd/topology/nodes/%d/gpu_id", i); cat(p, 1);
        snprintf(p, sizeof p, "/sys/class/kfd/kfd/topology/nodes/%d/properties", i); cat(p, 1);
        snprintf(p, sizeof p, "/sys/class/kfd/kfd/topology/nodes/%d/io_links/0/properties", i); cat(p, 1);
    }
    for (int i = 0; i < 9; ++i) { snprintf(p, sizeof p, "/sys/class/drm/renderD%d/device/numa_node", 128 + i); cat(p, 1); }
    { FILE *f = popen("ls -la /sys/bus/pci/devices/ | head -40; ls /sys/class/kfd/kfd/topology/nodes/", "r"); char l[512]; while (f && fgets(l, sizeof l, f)) fputs(l, stdout); if (f) pclose(f); }
    for (int i = 0; i < 3; ++i) { snprintf(p, sizeof p, "/sys/class/kfd/kfd/topology/nodes/%d/properties", 2); cat(p, 0); }
    return 0;
}
