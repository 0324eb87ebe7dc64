"""Registration estimation (mirror of ``biahub/registration``): intensity-based similarity estimate on the GPU."""
