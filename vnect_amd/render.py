"""Headless rendering of the results to image files (SURVEY 8 f-row 4; the reference: src/utils.py:222-330).

``draw_limbs_2d`` has the reference's signature and conventions (joints are [row, col]; limbs join joint i to
``limb_parents[i]``; the crop rectangle is outlined; colours are the reference's BGR triples) but draws with PIL, not
cv2: a limb is a 6-pixel-wide line instead of a filled 3-pixel half-axis ellipse, so pictures look alike without being
pixel-identical (there is no GUI here: results go to files).  ``draw_limbs_3d`` projects the skeleton orthographically
the way the reference's matplotlib view (elev -90, azim -90: x right, y down) shows it.
"""
import numpy as np

LIMB_BGR = (49, 22, 122)   # src/utils.py:238
RECT_BGR = (60, 66, 207)   # src/utils.py:243


def _canvas(img):
    from PIL import Image
    return Image.fromarray(np.ascontiguousarray(img[:, :, ::-1]))  # BGR -> RGB


def _back(pil):
    return np.ascontiguousarray(np.asarray(pil)[:, :, ::-1])


def draw_limbs_2d(img, joints_2d, limb_parents, rect):
    """Return a copy of the uint8 BGR frame with the skeleton and the crop rectangle drawn (src/utils.py:222-244)."""
    from PIL import ImageDraw
    pil = _canvas(img)
    d = ImageDraw.Draw(pil)
    for i, parent in enumerate(limb_parents):
        r1, c1 = joints_2d[i]
        r2, c2 = joints_2d[parent]
        d.line([(float(c1), float(r1)), (float(c2), float(r2))], fill=LIMB_BGR[::-1], width=6)
    x, y, w, h = rect
    d.rectangle([x, y, x + w, y + h], outline=RECT_BGR[::-1], width=4)
    return _back(pil)


def draw_limbs_3d(joints_3d, joint_parents, size=400, extent=500.0):
    """Orthographic front view of the 21x3 joints in mm (the reference's plot limits are +-500): uint8 BGR (size, size, 3)."""
    from PIL import Image, ImageDraw
    pil = Image.new("RGB", (size, size), (255, 255, 255))
    d = ImageDraw.Draw(pil)
    j = np.asarray(joints_3d, np.float64)
    px = (j[:, 0] + extent) / (2 * extent) * (size - 1)
    py = (j[:, 1] + extent) / (2 * extent) * (size - 1)
    for i, parent in enumerate(joint_parents):
        d.line([(px[i], py[i]), (px[parent], py[parent])], fill=(30, 30, 30), width=2)
    for i in range(len(j)):
        d.ellipse([px[i] - 3, py[i] - 3, px[i] + 3, py[i] + 3], fill=LIMB_BGR[::-1])
    return _back(pil)


def save(path, img_bgr):
    """cv2.imwrite replacement (PNG / JPEG by extension)."""
    _canvas(img_bgr).save(path)
