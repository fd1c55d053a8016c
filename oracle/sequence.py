"""numpy restatement of the reference's keyframe sequence loop (legacy/seq_example.py:150-208) on top of
banet_oracle.legacy_track  --  TEST INFRASTRUCTURE ONLY.  Point selection is shared with the product
(host logic, restated once in banet_amd/sequence.py from seq_example.py:72-83) by passing the points in."""
import numpy as np

from . import banet_oracle as orc


def run_sequence(intrinsics, frames, stamps, select, mlps, iters, min_keep_ratio=0.8, max_gap=0.1,
                 early_termination=True):
    """frames: list of per-frame pyramids (3 levels, each [1,H_l,W_l,C]); select(i) -> (points [1,N,2],
    depths [1,N,1]) for key frame i.  Returns the list of per-frame dict(rotation, translation, keep_ratio,
    globalRotation, globalTranslation, camera, new_keyframe, iters)."""
    dt = np.float32
    initR, initT = np.eye(3, dtype=dt)[None], np.zeros((1, 3, 1), dt)
    gRs, gTs = [initR], [initT]
    key = 0
    points, depths = select(0)
    out = []
    for i in range(1, len(frames)):
        layers = [np.concatenate([k, f], axis=0) for k, f in zip(frames[key], frames[i])]
        R, T, ratio, counts = orc.legacy_track(intrinsics, layers, points, depths, initR, initT, iters, mlps,
                                               early_termination=early_termination)
        ratio = float(np.squeeze(ratio))
        gR = np.matmul(R, gRs[key])
        gT = np.matmul(R, T) + gTs[key]
        gRs.append(gR)
        gTs.append(gT)
        camera = -np.matmul(np.transpose(gR, (0, 2, 1)).astype(np.float64), gT.astype(np.float64)).flatten()
        switched = ratio < min_keep_ratio or (float(stamps[i]) - float(stamps[key])) > max_gap
        out.append(dict(rotation=R, translation=T, keep_ratio=ratio, globalRotation=gR, globalTranslation=gT,
                        camera=camera, new_keyframe=switched, iters=counts))
        if switched:
            key = i
            points, depths = select(i)
            initR, initT = np.eye(3, dtype=dt)[None], np.zeros((1, 3, 1), dt)
        else:
            initR, initT = R, T
    return out
