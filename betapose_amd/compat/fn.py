"""``from fn import getTime, vis_frame, vis_frame_fast`` (fn.py:222-227; :88-220 are commented out in the reference)."""
import time


def getTime(time1=0):
    if not time1:
        return time.time()
    return time.time(), time.time() - time1


from betapose_amd.video import vis_frame, vis_frame_fast  # noqa: E402,F401
