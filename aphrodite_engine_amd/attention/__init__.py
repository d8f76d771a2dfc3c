from .backend import (MI355XAttentionBackend, MI355XAttentionImpl,  # noqa: F401
                      MI355XAttentionMetadata)
from .paged_attn import PagedAttention  # noqa: F401
