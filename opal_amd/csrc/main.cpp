// opal_amd <input.yaml>: the reference's command line (src/main.rs:55-61) on one GPU.
#include <cstdio>

#include "../../include/opal_amd.h"

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "Unable to open configuration file.\nUsage: opal_amd input-file\n");
        return 1;
    }
    opal_sim_t *sim = nullptr;
    int rc = opal_sim_create(&sim, argv[1], 0, 1, nullptr, nullptr);
    if (rc == OPAL_OK) rc = opal_sim_run(sim);
    if (rc != OPAL_OK) fprintf(stderr, "Error: %s\n", opal_last_error());
    opal_sim_free(sim);
    return rc == OPAL_OK ? 0 : 1;
}
