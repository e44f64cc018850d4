/*
 * sf_io.h — C ABI of the on-disk formats either side of the solver path:
 * the TUM-style association file + PNG frames the reference's image-sequence driver reads, and the
 * trajectory file its dataset driver writes. Host-only code (libsf_io.so, C++ + zlib): nothing here
 * touches the GPU; the decoded frames go to sf_load_frame() of sf.h.
 *
 *   reference                                                        here
 *   StaticFusion::loadAssoc            FrontEnd.cpp:183-214          sf_io_assoc_load / _count / _entry / _free
 *   cv::imread(rgb, CV_LOAD_IMAGE_COLOR) FrontEnd.cpp:220            sf_io_imread_color   (8-bit, 3 channels, B G R order)
 *   cv::imread(depth, -1)              FrontEnd.cpp:240              sf_io_imread_depth16 (16-bit, 1 channel, host byte order)
 *   currPose = currPose * T_odometry   Reconstruction.cpp:265        sf_io_pose_compose
 *   Datasets::writeTrajectoryFile      Utils/Datasets.cpp:252-265    sf_io_trajectory_line (rotate_by_z = 1)
 *   Reconstruction::~Reconstruction    Reconstruction.cpp:53-81      sf_io_trajectory_line (rotate_by_z = 0, fixed timestamp)
 *
 * Every function returns SF_IO_OK (0) or a negative code; sf_io_last_error() describes the last failure
 * of the calling thread.
 */
#ifndef SF_IO_H_
#define SF_IO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SF_IO_OK = 0, SF_IO_ERR_ARG = -1, SF_IO_ERR_FILE = -2, SF_IO_ERR_FORMAT = -3, SF_IO_ERR_UNSUPPORTED = -4 };

typedef struct sf_io_assoc sf_io_assoc;

/* loadAssoc(dir, assocFile, timestamps, filesDepth, filesColor): lines `ts_color file_color ts_depth file_depth`;
 * empty lines and lines starting with '#' are skipped; the first line that does not parse ends the list (the
 * reference `break`s, :204-205). timestamp = the DEPTH timestamp; paths = dir + file (plain concatenation, :208-209). */
int sf_io_assoc_load(const char *dir, const char *assoc_file, sf_io_assoc **out);
int sf_io_assoc_count(const sf_io_assoc *a);
int sf_io_assoc_entry(const sf_io_assoc *a, int index, double *timestamp, const char **depth_path, const char **color_path);
void sf_io_assoc_free(sf_io_assoc *a);

/* PNG -> what cv::imread(path, CV_LOAD_IMAGE_COLOR) returns: rows x cols x 3 uint8, interleaved B, G, R.
 * Grey images are replicated, alpha is dropped, 16-bit samples keep their high byte, palettes are expanded.
 * The buffer is malloc()ed; release with sf_io_free(). Adam7-interlaced files are SF_IO_ERR_UNSUPPORTED. */
int sf_io_imread_color(const char *path, uint8_t **bgr, int *rows, int *cols);
/* PNG -> what cv::imread(path, -1) returns for the reference's depth files (16-bit monochrome, "scaled by 1000",
 * reference README.md:84-87): rows x cols uint16 in host byte order. An 8-bit grey file is widened. */
int sf_io_imread_depth16(const char *path, uint16_t **depth, int *rows, int *cols);
/* the same decoders on an in-memory PNG */
int sf_io_decode_color(const uint8_t *png, size_t size, uint8_t **bgr, int *rows, int *cols);
int sf_io_decode_depth16(const uint8_t *png, size_t size, uint16_t **depth, int *rows, int *cols);
void sf_io_free(void *p);

/* out = pose * T (4x4, column-major float32, Eigen::Matrix4f storage; float accumulation in index order). */
void sf_io_pose_compose(const float pose[16], const float T[16], float out[16]);
/* One trajectory line `timestamp tx ty tz qx qy qz qw\n` into buf (NUL-terminated; returns the length, or a
 * negative code if buf is too small). rotate_by_z != 0: the pose is first multiplied by a rotation of pi about Z
 * and the timestamp printed with "%.04f" (Datasets.cpp:256-262); rotate_by_z == 0: timestamp with 6 fixed decimals
 * (Reconstruction.cpp:66). Floats are printed as std::ostream does by default (6 significant digits); the
 * quaternion follows Eigen::Quaternionf(Matrix3f). */
int sf_io_trajectory_line(double timestamp, const float pose[16], int rotate_by_z, char *buf, size_t buf_size);

/* Reconstruction::savePly (reference Reconstruction.cpp:358-455), the map half: binary little-endian PLY of the surfels
 * whose confidence exceeds conf_threshold -- x y z (float), red green blue (uchar, from the encoded colour), nx ny nz
 * (float, NEGATED as the reference does), radius (float). surfels: count x 12 floats in the global model's vertex
 * layout (sf_map_download). Returns the number of vertices written or a negative code. */
int sf_io_save_ply(const char *path, const float *surfels, int count, float conf_threshold);

const char *sf_io_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SF_IO_H_ */
