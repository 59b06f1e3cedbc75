"""Offline pipeline: dataset access, hidden-state / LM-head extraction."""
