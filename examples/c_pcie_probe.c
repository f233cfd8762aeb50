/* c_pcie_probe.c -- pcx_pcie_probe from a plain C process (the system's HIP runtime, no Python, no torch):
 * the PCIe roof of the box as the copy engines see it.  Build: make -C examples */
#include <stdio.h>

#include "pcx.h"

int main(void)
{
    double up = 0, down = 0, both = 0;
    if (pcx_pcie_probe((size_t)128 << 20, 5, &up, &down, &both) != PCX_OK) {
        fprintf(stderr, "pcx_pcie_probe: %s\n", pcx_last_error());
        return 1;
    }
    printf("plain C process: H2D alone %.1f GB/s, D2H alone %.1f, H2D || D2H on two streams %.1f per direction\n", up, down, both);
    return 0;
}
