"""diff_sal_amd -- MI355X-native DiffSal denoising hot path (SalUNet, samplers, training step) behind libdiffsal_hip.so."""
from .sal_unet import SalUNet  # noqa: F401
from .diff_model import VideoSaliencyModel  # noqa: F401
from .mvit import MViT  # noqa: F401
from .vggish import VGGish  # noqa: F401
from .audio_attention import AudioAttnNet  # noqa: F401
from .diffusion_unet import DiffusionModel, DiffusionModel_w_MultiScale  # noqa: F401
from .ema import EMAHelper  # noqa: F401
from .dpm_solver import DPM_Solver, NoiseScheduleVP, model_wrapper  # noqa: F401
from .sampling import DiffusionSampler, ddpm_steps, generalized_steps  # noqa: F401
from .diffusion_utils import get_beta_schedule, to_torch  # noqa: F401
from .train_step import DiffusionTrainStep, FlatParams, GradReducer  # noqa: F401

__all__ = ["SalUNet", "VideoSaliencyModel", "MViT", "VGGish", "AudioAttnNet", "DiffusionModel", "DiffusionModel_w_MultiScale", "EMAHelper", "DPM_Solver", "NoiseScheduleVP", "model_wrapper", "DiffusionSampler",
           "generalized_steps", "ddpm_steps", "get_beta_schedule", "to_torch", "DiffusionTrainStep", "FlatParams", "GradReducer"]
