"""Test-only stand-in for the ``openslide`` package (f4): an OpenSlide-protocol reader with a multi-level pyramid.

The reference reads every tile as ``slide.read_region((x, y) in level-0 pixels, level, (w, h))`` -> RGBA ``PIL.Image`` ->
``np.array`` -> drop alpha, at the pyramid level ``get_best_level_for_downsample(train_mpp / mpp)`` picks
(/root/reference/src/classpose/entrypoints/predict_wsi.py:220-278, 446-451), through the reader class its
``WSI_READER`` switch imports (``from openslide import OpenSlide``, src/classpose/__init__.py:6-41).  There is no
OpenSlide wheel (and no slide file) in the build image, so the tests install THIS module as ``sys.modules["openslide"]``:
same constructor, attributes and return types as openslide-python 1.4, pixels from the procedural slide of
``classpose_amd.synth`` placed at ONE pyramid level (``base_level``), so that a run over this reader can be compared
with the ``synthetic://`` run of the same pixels.

Also a plug-in (``CLASSPOSE_AMD_PLUGINS=fake_openslide``): the flow-injection fields of tiles read through this reader.
"""
from __future__ import annotations

import numpy as np
from PIL import Image

from classpose_amd import synth

SLIDES: dict[str, dict] = {}          # path -> spec, filled by the test before the CLI opens the path
READS: list[tuple] = []               # (path, location, level, size) of every read_region call


class OpenSlideError(Exception):
    pass


class OpenSlide:
    def __init__(self, filename):
        self._filename = str(filename)
        if self._filename not in SLIDES:
            raise OpenSlideError(f"Unsupported or missing image file: {self._filename}")
        spec = SLIDES[self._filename]
        self.seed = int(spec["seed"])
        self.base_level = int(spec["base_level"])
        self.level_downsamples = tuple(float(d) for d in spec["downsamples"])
        bw, bh = spec["base_dims"]
        db = self.level_downsamples[self.base_level]
        self.level_dimensions = tuple((int(round(bw * db / d)), int(round(bh * db / d))) for d in self.level_downsamples)
        self.level_count = len(self.level_downsamples)
        self.dimensions = self.level_dimensions[0]
        self.properties = dict(spec["properties"])            # str -> str, like OpenSlide's property map

    def get_best_level_for_downsample(self, downsample: float) -> int:
        """openslide_get_best_level_for_downsample: the level with the largest downsample <= the requested one
        (level 0 when the request is below every level)."""
        best = 0
        for i, d in enumerate(self.level_downsamples):
            if d <= downsample:
                best = i
        return best

    def read_region(self, location, level: int, size) -> Image.Image:
        READS.append((self._filename, (int(location[0]), int(location[1])), int(level), (int(size[0]), int(size[1]))))
        if level != self.base_level:
            raise OpenSlideError(f"fake slide holds pixels at level {self.base_level} only, level {level} was read")
        d = self.level_downsamples[level]
        x, y = location[0] / d, location[1] / d             # location is in the level-0 frame
        if x != int(x) or y != int(y):
            raise OpenSlideError(f"read origin {location} does not sit on the level-{level} pixel grid")
        w, h = int(size[0]), int(size[1])
        rgb = synth.render_region(self.seed, int(x), int(y), w, h)
        rgba = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=-1)
        return Image.fromarray(rgba, "RGBA")

    def get_thumbnail(self, size) -> Image.Image:
        w, h = int(size[0]), int(size[1])
        bw, bh = self.level_dimensions[self.base_level]
        xs = (np.arange(w) * (bw / w)).astype(np.int64)
        ys = (np.arange(h) * (bh / h)).astype(np.int64)
        return Image.fromarray(synth.render_points(self.seed, xs, ys), "RGB")

    def close(self) -> None:
        pass


def register(hooks, argument: str) -> None:
    """flow-injection fields for tiles of the fake reader: the procedural nuclei live on the base level's pixel grid, the
    tile origins the CLI hands over are level-0 pixels"""
    def field_provider(slide, plan, n_classes):
        if isinstance(slide, OpenSlide):
            d = slide.level_downsamples[plan.level]
            return lambda ti, R, W, H: synth.analytic_fields(slide.seed, int(plan.coords[ti][0][0] / d), int(plan.coords[ti][0][1] / d),
                                                             R, R, n_classes, W, H)
        if hasattr(slide, "seed"):
            return lambda ti, R, W, H: synth.analytic_fields(slide.seed, plan.coords[ti][0][0], plan.coords[ti][0][1], R, R,
                                                             n_classes, W, H)
        return None
    hooks.field_provider = field_provider
