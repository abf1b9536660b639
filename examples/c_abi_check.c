/* The boundary is a plain-C ABI: this file is compiled as C (gcc -std=c99 -pedantic) against include/betapose_hip.h and
 * linked with libbetapose_hip.so.  It needs no GPU: it checks the library answers, reports the device count and fails
 * cleanly (status < 0 + message) when asked for an engine it cannot build.
 *   gcc -std=c99 -pedantic -Wall -Werror -Iinclude examples/c_abi_check.c -o /tmp/c_abi_check \
 *       -Lbetapose_amd -lbetapose_hip -Wl,-rpath,$PWD/betapose_amd */
#include <stdio.h>
#include <string.h>

#include "betapose_hip.h"

int main(void) {
    bp_yolo* y = NULL;
    double R[9], t[3];
    /* a known-answer PnP: identity rotation, t = (0, 0, 1), K = LineMod intrinsics; no device involved */
    const double K[9] = {572.4114, 0.0, 325.2611, 0.0, 573.57043, 242.04899, 0.0, 0.0, 1.0};
    const double P[8][3] = {{-.05, -.05, -.05}, {.05, -.05, -.04}, {-.05, .05, .03}, {.05, .05, -.02},
                            {-.03, .01, .05},   {.02, -.04, .04},  {.04, .03, .01},  {-.01, -.02, -.03}};
    double p2[8][2];
    int i, rc;
    printf("bp_version %d, devices %d\n", bp_version(), bp_device_count());
    if (bp_version() < 100) return 1;
    for (i = 0; i < 8; ++i) {
        const double z = P[i][2] + 1.0;
        p2[i][0] = K[0] * P[i][0] / z + K[2];
        p2[i][1] = K[4] * P[i][1] / z + K[5];
    }
    rc = bp_solve_pnp(&P[0][0], &p2[0][0], 8, K, R, t);
    if (rc != 0 || t[2] < 0.999 || t[2] > 1.001 || R[0] < 0.9999) {
        printf("bp_solve_pnp: rc %d t_z %f R00 %f (%s)\n", rc, t[2], R[0], bp_last_error());
        return 2;
    }
    rc = bp_yolo_create("/nonexistent.cfg", "/nonexistent.weights", 416, 1, 0, &y);
    if (rc >= 0 || y != NULL || strlen(bp_last_error()) == 0) return 3;
    printf("expected failure reported: %s\n", bp_last_error());
    return 0;
}
