"""d3d_amd.tracking -- the association step of d3d.tracking (reference d3d/tracking/matcher.pyx) on MI355X."""
from .matcher import DistanceTypes, ScoreMatcher, prepare_boxes, score_match, score_match_reference_compat

__all__ = ["DistanceTypes", "ScoreMatcher", "prepare_boxes", "score_match", "score_match_reference_compat"]
