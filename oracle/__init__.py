"""Test infrastructure only: CPU oracle of the reference hot path. Never imported by composer_amd."""
