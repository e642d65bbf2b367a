from .att_pooling import GlobalAttention  # noqa: F401
from .build import build_model  # noqa: F401
from .isubgvqa import ISubGVQA  # noqa: F401
from .masking import MaskingModel, get_aimle_samplers, get_imle_samplers  # noqa: F401
from .mgat import MGAT  # noqa: F401
from .mgat_v2_conv import MaskingGATv2Conv  # noqa: F401
from .scene_graph_encoder import SceneGraphEncoder  # noqa: F401
from .text_encoder import CLIPTextEmbeddings, QuestionDecoder, QuestionEncoder  # noqa: F401
