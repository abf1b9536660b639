/* The six extern "C" entry points and the structs of the reference's Darknet detector library
 * (3_6Dpose_estimator/train_YOLO/src/yolo_v2_class.hpp:16-54), served by libbetapose_hip.so on an MI355X.
 * Same names, argument meaning and return conventions: init -> 1; detect_* -> number of detections (at most
 * C_SHARP_MAX_OBJECTS are written), negative on error (the reference's detect_mat returns -1 when built without
 * OpenCV); dispose -> 1; get_device_name -> 1 on success.  Detection threshold 0.2 and NMS 0.4 as in the reference's
 * Detector.  Differences: image files are PNG, baseline / progressive JPEG or uncompressed BMP (the formats the
 * reference's stb_image loads that this library does not: GIF, TGA, PSD, HDR, PIC, PNM); the network input must be square.  C++ header (the
 * container is passed by reference, as in the reference). */
#ifndef YOLO_V2_CLASS_COMPAT_H
#define YOLO_V2_CLASS_COMPAT_H
#include <stddef.h>
#include <stdint.h>

#define C_SHARP_MAX_OBJECTS 1000

struct bbox_t {
    unsigned int x, y, w, h;       /* (x, y) top-left corner, (w, h) size, image pixels */
    float prob;                    /* objectness * class probability */
    unsigned int obj_id;           /* class id */
    unsigned int track_id;         /* always 0 here */
    unsigned int frames_counter;   /* always 0 here */
};

struct bbox_t_container {
    bbox_t candidates[C_SHARP_MAX_OBJECTS];
};

extern "C" int init(const char* configurationFilename, const char* weightsFilename, int gpu);
extern "C" int detect_image(const char* filename, bbox_t_container& container);
extern "C" int detect_mat(const uint8_t* data, const size_t data_length, bbox_t_container& container);
extern "C" int dispose();
extern "C" int get_device_count();
extern "C" int get_device_name(int gpu, char* deviceName);

#endif /* YOLO_V2_CLASS_COMPAT_H */
