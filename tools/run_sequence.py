"""The reference's image-sequence driver (StaticFusion-imagesequenceassoc.cpp:57-191) over the C ABIs of this
repository. --mode frame: the prediction is the previous (filtered) frame (mode (a)).
--mode keyframe: frame-to-model against the surfel model of the first frame. --mode fusion: the reference's full loop
without OpenGL -- every solved frame is fused into the surfel map (sf_map_fuse_frame) and the next prediction is
rendered from it (sf_map_predict).

  dataset/rgb/*.png  dataset/depth/*.png  dataset/rgbd_assoc.txt      (reference README.md:67-89)

usage: python tools/run_sequence.py <dataset dir> [--out trajectory.txt] [--max-frames N] [--res-factor 2]
Writes one `timestamp tx ty tz qx qy qz qw` line per frame (the .freiburg format of Reconstruction.cpp:53-81).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(api, io, directory, assoc_file="rgbd_assoc.txt", res_factor=2, max_frames=None, out_path=None, filter_first=False, mode="frame", prediction_poses=None,
        odometry_override=None, capacity=0, trace=None):
    """api: an include/sf.h implementation (staticfusion_amd.load() or the test oracle); io: staticfusion_amd.io.Io().
    mode "frame": the prediction is the previous filtered frame. mode "keyframe": frame-to-MODEL tracking against the
    surfel model of the first frame (GlobalModel::initialise, never fused further): every prediction is rendered from
    it at the current pose (Reconstruction::getPredictedImages), as the reference does with its growing map.
    prediction_poses: render the predictions at THESE poses instead of the own estimate (a visibility test flips pixels
    for a 1e-7 change of the pose, so two implementations are compared frame by frame on identical predictions).
    mode "fusion": StaticFusion-imagesequenceassoc.cpp:102-186 in full: frame 0 becomes the first prediction, frame 1 is
    solved against it and initialises the map (fuseFrame with tick 1), every later frame is predicted from the map, solved,
    and fused. odometry_override[k]: fuse frame k with THIS T_odometry instead of the own estimate (teacher forcing for
    parity runs: the own estimate is still what `increments` reports). trace: a list that receives one dict per frame."""
    import staticfusion_amd as sf

    if not directory.endswith("/"):
        directory += "/"
    ts, files_depth, files_color = io.load_assoc(directory, assoc_file)  # loadAssoc, :96
    if max_frames is not None:
        ts, files_depth, files_color = ts[:max_frames], files_depth[:max_frames], files_color[:max_frames]
    if not ts:
        raise RuntimeError("empty association file")
    color, depth = io.imread_color(files_color[0]), io.imread_depth16(files_depth[0])
    rows, cols = depth.shape[0] // res_factor, depth.shape[1] // res_factor
    p = api.default_params_struct()  # the drivers' parameter block, StaticFusion-imagesequenceassoc.cpp:62-79
    s = sf.Solver(api, rows, cols, 1, p)
    pose = np.eye(4, dtype=np.float32)
    poses, lines = [pose.copy()], [io.trajectory_line(ts[0], pose, 0)]
    # bootstrap (:102-137): the first frame becomes the prediction; kb = 1.05 until the model is dense (:152-163) --
    # without a map, every frame is "not dense": kb stays 1.05
    s.load_frame(0, color, depth, res_factor)
    if filter_first:
        s.filter_depth()
    s.current_to_prediction()
    s.push_history(0)
    s.set_kb(1.05)
    if mode == "fusion":
        return _run_fusion(sf, api, io, s, ts, files_depth, files_color, res_factor, out_path, odometry_override, capacity, trace)
    model = mp = None
    if mode == "keyframe":
        s.filter_depth()
        mp = s.default_model_params()
        model = s.init_model_from_frame(0, pose, mp, time=1)  # fuseFrame of the bootstrap frame (imagesequenceassoc.cpp:135)
    for k in range(1, len(ts)):
        if model is not None:  # getPredictedImages BEFORE the new frame is loaded (:164)
            mp.time = mp.max_time = k + 1
            s.predict_from_model(0, model, pose if prediction_poses is None else prediction_poses[k - 1], mp)
        color, depth = io.imread_color(files_color[k]), io.imread_depth16(files_depth[k])  # loadImageFromSequenceAssoc, :149
        s.load_frame(0, color, depth, res_factor)
        s.filter_depth()      # reconstruction->getFilteredDepth(depth_mm, depthCurrent), :165
        s.process_frame(k)    # createImagePyramid(true); runSolver(true); residuals (k >= 5); buildSegmImage; ring push, :167-179
        pose = io.pose_compose(pose, s.T())  # currPose = currPose * T_odometry, Reconstruction.cpp:265
        poses.append(pose.copy())
        lines.append(io.trajectory_line(ts[k], pose, 0))
        if model is None:
            s.current_to_prediction()  # frame-to-frame mode
    if out_path:
        with open(out_path, "w") as f:
            f.writelines(lines)
    return poses, lines, s


def _run_fusion(sf, api, io, s, ts, files_depth, files_color, res_factor, out_path, odometry_override, capacity, trace):
    m = sf.SurfelMap(s, capacity)
    mp = s.default_model_params()
    pose = np.eye(4, dtype=np.float32)
    poses, lines, increments = [pose.copy()], [io.trajectory_line(ts[0], pose, 0)], [np.eye(4, dtype=np.float32)]
    for k in range(1, len(ts)):
        if k >= 2:
            s.set_kb(1.05 if k == 2 else 1.5)  # :151-163: checkIfDenseEnough reads the low-confidence image of the PREVIOUS
            m.predict(0, mp)                   # getPredictedImages; nothing has been rendered before the first one. :164
        color, depth = io.imread_color(files_color[k]), io.imread_depth16(files_depth[k])
        s.load_frame(0, color, depth, res_factor)
        if k >= 2:
            s.filter_depth()                   # getFilteredDepth(depth_mm, depthCurrent), :165 (the bootstrap frame is solved unfiltered, :117-123)
        s.process_frame(k)                     # pyramid, runSolver, residuals, buildSegmImage, ring push, :119-130 / :167-179
        T = s.T()
        increments.append(T.copy())
        if k == 1:
            s.filter_depth()                   # fuseFrame filters the depth it is given itself (Reconstruction.cpp:246-247)
        m.fuse_frame(0, T if odometry_override is None else odometry_override[k], 1.0, mp)  # :135 / :183
        info = m.info()
        pose = info["pose"]
        poses.append(pose.copy())
        lines.append(io.trajectory_line(ts[k], pose, 0))
        if trace is not None:
            trace.append(dict(frame=k, count=info["count"], stats=info["stats"], T=T.copy()))
    if out_path:
        with open(out_path, "w") as f:
            f.writelines(lines)
    s.map, s.increments = m, increments
    return poses, lines, s


def main():
    import staticfusion_amd as sf
    from staticfusion_amd import io as sfio

    ap = argparse.ArgumentParser()
    ap.add_argument("dataset")
    ap.add_argument("--assoc", default="rgbd_assoc.txt")
    ap.add_argument("--out", default="trajectory.freiburg")
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("--res-factor", type=int, default=2)
    ap.add_argument("--mode", choices=["frame", "keyframe", "fusion"], default="frame")
    a = ap.parse_args()
    if not os.path.isdir(a.dataset):
        print("dataset absent: %s (no datasets ship with this repository; see SURVEY.md §8(d) config 1)" % a.dataset)
        return 2
    poses, lines, _ = run(sf.load(), sfio.Io(), a.dataset, a.assoc, a.res_factor, a.max_frames, a.out, mode=a.mode)
    print("%d frames -> %s" % (len(poses), a.out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
