"""Evaluation: acceptance metrics, two-phase evaluator, timers, profilers, reports."""
from .acceptance import (SDConfig, TimingConfig, compute_all_metrics_parallel,  # noqa: F401
                         compute_cosim_vs_token_correlation, compute_per_position_stats,
                         compute_token_level_metrics, compute_two_phase_sd_metrics, project_to_tokens)
