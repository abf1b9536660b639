// A consumer of the reference's Darknet detector library compiles unchanged against include/yolo_v2_class_compat.h
// (g++ -std=c++11) and links libbetapose_hip.so.  Without init() the calls fail cleanly.
#include <cstdio>

#include "yolo_v2_class_compat.h"

int main() {
    static bbox_t_container boxes;
    static_assert(sizeof(bbox_t) == 32, "bbox_t layout (yolo_v2_class.hpp:16-22)");
    static_assert(sizeof(bbox_t_container) == 32 * C_SHARP_MAX_OBJECTS, "container layout");
    if (detect_image("none.png", boxes) >= 0) return 1;       // init() not called
    if (dispose() != 1) return 2;
    std::printf("devices: %d\n", get_device_count());
    return 0;
}
