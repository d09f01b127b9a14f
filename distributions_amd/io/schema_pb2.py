"""Protobuf message classes of the reference's wire format, built at import
time from a field table (this image has no protoc).

Restates the messages of distributions/io/schema.proto that lie either side
of the row-update path (package `protobuf.distributions`): the clustering
hyper-parameters and, per component model, `Shared` (hyper-parameters) and
`Group` (sufficient statistics).  Message and field names, field numbers,
types and labels are the interface; tests/golden/schema_fields.json holds the
same table extracted from the descriptor embedded in the reference's own
generated module, and tests/test_io.py compares the two and checks serialized
bytes against messages built from the reference's descriptor.

Usage is that of the reference's module:

    from distributions_amd.io import schema_pb2
    message = schema_pb2.DirichletDiscrete.Group()
    group.protobuf_dump(message)
    data = message.SerializeToString()
"""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

PACKAGE = 'protobuf.distributions'

_F = descriptor_pb2.FieldDescriptorProto
_TYPES = {
    'float': _F.TYPE_FLOAT, 'uint64': _F.TYPE_UINT64,
    'uint32': _F.TYPE_UINT32, 'int32': _F.TYPE_INT32,
}
_LABELS = {
    'required': _F.LABEL_REQUIRED, 'optional': _F.LABEL_OPTIONAL,
    'repeated': _F.LABEL_REPEATED,
}

# message -> [(label, type, name, number)]; a type that is not scalar names a
# nested message of the same outer message
SCHEMA = {
    'Clustering': [
        ('optional', 'PitmanYor', 'pitman_yor', 1),
        ('optional', 'LowEntropy', 'low_entropy', 2),
    ],
    'Clustering.PitmanYor': [
        ('required', 'float', 'alpha', 1),
        ('required', 'float', 'd', 2),
    ],
    'Clustering.LowEntropy': [
        ('required', 'uint64', 'dataset_size', 1),
    ],
    'BetaBernoulli.Shared': [
        ('required', 'float', 'alpha', 1),
        ('required', 'float', 'beta', 2),
    ],
    'BetaBernoulli.Group': [
        ('required', 'uint64', 'heads', 1),
        ('required', 'uint64', 'tails', 2),
    ],
    'DirichletDiscrete.Shared': [
        ('repeated', 'float', 'alphas', 1),
    ],
    'DirichletDiscrete.Group': [
        ('repeated', 'uint64', 'counts', 1),
    ],
    'DirichletProcessDiscrete.Shared': [
        ('required', 'float', 'gamma', 1),
        ('required', 'float', 'alpha', 2),
        ('repeated', 'uint32', 'values', 3),
        ('repeated', 'float', 'betas', 4),
        ('repeated', 'uint64', 'counts', 5),
    ],
    'DirichletProcessDiscrete.Group': [
        ('repeated', 'uint32', 'keys', 1),
        ('repeated', 'uint64', 'values', 2),
    ],
    'PitmanYorProcessDiscrete.Shared': [
        ('required', 'float', 'alpha', 1),
        ('repeated', 'float', 'd', 2),
        ('repeated', 'uint64', 'counts', 3),
    ],
    'PitmanYorProcessDiscrete.Group': [
        ('repeated', 'uint32', 'keys', 1),
        ('repeated', 'uint64', 'values', 2),
    ],
    'GammaPoisson.Shared': [
        ('required', 'float', 'alpha', 1),
        ('required', 'float', 'inv_beta', 2),
    ],
    'GammaPoisson.Group': [
        ('required', 'uint64', 'count', 1),
        ('required', 'uint64', 'sum', 2),
        ('required', 'float', 'log_prod', 3),
    ],
    'BetaNegativeBinomial.Shared': [
        ('required', 'float', 'alpha', 1),
        ('required', 'float', 'beta', 2),
        ('required', 'uint64', 'r', 3),
    ],
    'BetaNegativeBinomial.Group': [
        ('required', 'uint64', 'count', 1),
        ('required', 'uint64', 'sum', 2),
    ],
    'NormalInverseChiSq.Shared': [
        ('required', 'float', 'mu', 1),
        ('required', 'float', 'kappa', 2),
        ('required', 'float', 'sigmasq', 3),
        ('required', 'float', 'nu', 4),
    ],
    'NormalInverseChiSq.Group': [
        ('required', 'uint64', 'count', 1),
        ('required', 'float', 'mean', 2),
        ('required', 'float', 'count_times_variance', 3),
    ],
    'NormalInverseWishart.Shared': [
        ('repeated', 'float', 'mu', 1),
        ('required', 'float', 'kappa', 2),
        ('repeated', 'float', 'psi', 3),
        ('required', 'float', 'nu', 4),
    ],
    'NormalInverseWishart.Group': [
        ('required', 'int32', 'count', 1),
        ('repeated', 'float', 'sum_x', 2),
        ('repeated', 'float', 'sum_xxT', 3),
    ],
}


def _build_file():
    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.name = 'distributions_amd/io/schema.proto'
    fdp.package = PACKAGE
    fdp.syntax = 'proto2'
    protos = {}
    for full in SCHEMA:             # outer messages first (dict order)
        parts = full.split('.')
        outer = parts[0]
        if outer not in protos:
            protos[outer] = fdp.message_type.add()
            protos[outer].name = outer
        if len(parts) == 2 and full not in protos:
            protos[full] = protos[outer].nested_type.add()
            protos[full].name = parts[1]
    for full, fields in SCHEMA.items():
        outer = full.split('.')[0]
        for label, typ, name, number in fields:
            f = protos[full].field.add()
            f.name = name
            f.number = number
            f.label = _LABELS[label]
            if typ in _TYPES:
                f.type = _TYPES[typ]
            else:
                f.type = _F.TYPE_MESSAGE
                f.type_name = '.%s.%s.%s' % (PACKAGE, outer, typ)
    return fdp


FILE = _build_file()
_pool = descriptor_pool.DescriptorPool()
_pool.Add(FILE)


def _message_class(full):
    return message_factory.GetMessageClass(
        _pool.FindMessageTypeByName('%s.%s' % (PACKAGE, full)))


for _outer in sorted({full.split('.')[0] for full in SCHEMA}):
    globals()[_outer] = _message_class(_outer)
del _outer
