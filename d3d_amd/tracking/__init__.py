"""d3d_amd.tracking -- the association step of d3d.tracking (reference d3d/tracking/matcher.pyx) on MI355X."""
from .matcher import DistanceTypes, ScoreMatcher, prepare_boxes, score_match

__all__ = ["DistanceTypes", "ScoreMatcher", "prepare_boxes", "score_match"]
